"""Device-side TLAS maintenance (hrt_scene_update_instances = BvhManager.BuildOrRefit with the
RebuildPolicy the reference declares and ignores, BvhManager.cs:13-27).

What is checked, all bit-exact:
  * instance records re-derived on the device == the oracle's restatement of InvertRigidOrUniform / TransformAABB;
  * a refit == the same topology with every box recomputed bottom-up (numpy restatement below), a refit without any
    move == the uploaded tree;
  * a rebuilt tree is a valid TLAS (every instance in exactly one leaf, <= 2 per leaf, every box the union of its
    children, every node on the walk) and the same moves give the same tree again;
  * frames rendered on the device-maintained tree == the oracle rendering the DOWNLOADED arrays, in every kernel
    organisation, and == the frame of the same scene rebuilt on the host wherever the reference's picture does not depend on the
    tree (translated / shrunk instances, no exact hit-distance ties in these scenes)."""
import ctypes as C

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu

CFG_SPH = scenes.Config("upd", 0, 0, 0, (0.0, 3.0, 9.0), (0.0, 0.8, 0.0))
CFG_ROT = scenes.Config("rot", 0, 0, 0, (0.4, 1.6, 4.6), (0.0, 0.8, 0.0))


def _spheres(b):
    scenes.build_random_spheres(b, 60, extent=4.0)


SCENES = {
    "sphere_instances": (_spheres, CFG_SPH, 128, 80, 2),
    "cornell_flat_leaves": (scenes.build_config2, scenes.CONFIGS[2], 128, 72, 2),
    "rotated_mesh_and_sets": (scenes.build_rotated_instances_scene, CFG_ROT, 128, 80, 2),
}
MODES = {"auto": T.FLAG_COUNTERS, "auto_production": 0, "stream_production": T.FLAG_STREAMED, "mega_production": T.FLAG_MEGAKERNEL,
         "stream_reflayout": T.FLAG_COUNTERS | T.FLAG_STREAMED | T.FLAG_REFERENCE_LAYOUT}


def _moves(n_inst, kind):
    """instance ids + transforms.  kind: 'rigid' = rotation, uniform scale up to 1.2 and translation; 'translate' =
    translation with a uniform scale <= 1; 'identity' = nothing really moves."""
    ids = [i for i in range(n_inst) if i % 3 != 0][:24]          # instance 0 (the ground) stays
    xfs = []
    for k, i in enumerate(ids):
        if kind == "identity":
            xfs.append(T.identity_affine())
        elif kind == "hostile":      # zero / negative / huge uniform scale, a NaN and an infinite entry, a shear (columns of unequal length)
            m = scenes.rotation_affine("xyz"[k % 3], 23.0 * k, [0.0, -1.3, 1e19, 1.0, 0.6, 1.0][k % 6], (0.2 * (k % 4) - 0.3, 0.1 * (k % 3), 0.2 * (k % 5) - 0.4))
            if k % 6 == 3: m.m12 = float("nan")
            if k % 6 == 4: m.m03 = float("inf")
            if k % 6 == 5: m.m01 = 0.7; m.m22 = 2.5
            xfs.append(m)
        elif kind == "translate":
            xfs.append(scenes.rotation_affine("y", 0.0, 1.0 if k % 3 else 0.75, (0.3 * (k % 4) - 0.4, 0.1 * (k % 3), 0.25 * (k % 5) - 0.5)))
        elif k % 2 == 0:
            xfs.append(scenes.rotation_affine("xyz"[k % 3], 17.0 * k, 1.0 + 0.05 * (k % 5), (0.3 * (k % 4) - 0.4, 0.1 * (k % 3), 0.25 * (k % 5) - 0.5)))
        else:
            xfs.append(scenes.rotation_affine("y", 0.0, 1.0, (0.15 * k - 1.0, 0.05 * k, -0.1 * k)))
    return ids, xfs


def _as_np(ct_array, n, t):
    return np.frombuffer(ct_array, dtype=T.np_dtype(t), count=n).copy()


def _download(r):
    nodes, idx, inst, cnt = r.download_tlas()
    return _as_np(nodes, cnt[0], T.BvhNode), np.frombuffer(idx, dtype=np.int32, count=cnt[1]).copy(), _as_np(inst, cnt[2], T.InstanceRecord)


def _walk(nodes, idx):
    """The tree as the walk sees it: [(boundsMin, boundsMax, tuple(instances of a leaf) or None)] in walk order."""
    out, cur, guard = [], 0 if len(nodes) else -1, 0
    nodes = H.canon(nodes)                      # a NaN box (hostile moves) is "a NaN box" on both machines
    while cur != -1:
        n = nodes[cur]
        leaf = n["count"] > 0
        out.append((n["boundsMin"].tobytes(), n["boundsMax"].tobytes(), tuple(int(v) for v in idx[n["first"]:n["first"] + n["count"]]) if leaf else None))
        cur = int(n["skipIndex"]) if leaf else int(n["left"])
        guard += 1
        assert guard <= len(nodes), "walk does not end"
    return out


# Every union of the builders starts from the inverted float.MaxValue box (Scene.cs:386-390,472-480,560-580), which matters for
# infinite bounds only: Min(float.MaxValue, +inf) = float.MaxValue.
FMAX3 = np.full(3, np.finfo(np.float32).max, np.float32)


def _refit_numpy(nodes, idx, inst):
    """Same topology, every box recomputed: leaves from their instances' world bounds, inner nodes from their children
    (left, then the skip chain up to the node's own skip link)."""
    nodes = nodes.copy()

    def rec(i):
        n = nodes[i]
        if n["count"] > 0:
            ids = idx[n["first"]:n["first"] + n["count"]]
            lo = np.min(np.stack([FMAX3] + [np.array(list(inst[j]["worldBoundsMin"].tolist()), np.float32) for j in ids]), axis=0)
            hi = np.max(np.stack([-FMAX3] + [np.array(list(inst[j]["worldBoundsMax"].tolist()), np.float32) for j in ids]), axis=0)
        else:
            los, his, c = [], [], int(n["left"])
            while c != -1 and c != int(n["skipIndex"]):
                a, b = rec(c)
                los.append(a); his.append(b)
                c = int(nodes[c]["skipIndex"])
            lo, hi = np.min(np.stack([FMAX3] + los), axis=0), np.max(np.stack([-FMAX3] + his), axis=0)
        for k, f in enumerate("XYZ"):
            nodes[i]["boundsMin"][f] = lo[k]
            nodes[i]["boundsMax"][f] = hi[k]
        return lo, hi

    if len(nodes):
        rec(0)
    return nodes


def _desc_with_tlas(base_desc, nodes, idx, inst):
    """base_desc with the three TLAS-side arrays replaced (numpy structured arrays kept alive by the caller)."""
    d = T.SceneDesc()
    C.memmove(C.byref(d), C.byref(base_desc), C.sizeof(d))
    d.tlasNodes = C.cast(nodes.ctypes.data, C.POINTER(T.BvhNode)); d.n_tlasNodes = len(nodes)
    d.tlasInstanceIndices = C.cast(idx.ctypes.data, C.POINTER(C.c_int32)); d.n_tlasInstanceIndices = len(idx)
    d.instances = C.cast(inst.ctypes.data, C.POINTER(T.InstanceRecord)); d.n_instances = len(inst)
    return d


def _oracle_render(orc, desc, cfg, w, h, spp):
    p = scenes.frame_params(cfg, *H.host_funcs("orc", orc), width=w, height=h, spp=spp)
    arrs, o = T.alloc_outputs(w, h)
    st = orc.render_frame(desc, p, o, None)
    return arrs, st


def _gpu_render(r, cfg, w, h, spp, flags):
    r.reset_history()
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
    arrs, o = T.alloc_outputs(w, h)
    st = r.render_params(p, o, flags=flags)
    return arrs, st


def _check_frames(orc, r, desc, cfg, w, h, spp):
    ref, ost = _oracle_render(orc, desc, cfg, w, h, spp)
    for mode, flags in MODES.items():
        got, gst = _gpu_render(r, cfg, w, h, spp, flags)
        H.assert_outputs_equal(ref, got)
        if flags & T.FLAG_COUNTERS:
            for i in range(2):
                assert gst.k[i].as_dict() == ost.k[i].as_dict(), (mode, i)
    return ref


def _check_valid_tlas(nodes, idx, inst):
    n_inst = len(inst)
    assert sorted(idx.tolist()) == list(range(n_inst)), "every instance in exactly one leaf slot"
    walk = _walk(nodes, idx)
    assert len(walk) == len(nodes), "every node is on the walk"
    leaves = sum(1 for _, _, l in walk if l is not None)
    assert len(nodes) == 2 * leaves - 1 and (n_inst + 1) // 2 <= leaves <= n_inst
    assert all(1 <= len(l) <= 2 for _, _, l in walk if l is not None)
    assert sum(len(l) for _, _, l in walk if l is not None) == n_inst
    again = _refit_numpy(nodes, idx, inst)
    assert H.canon(nodes).tobytes() == H.canon(again).tobytes(), "every box is the union of what it holds"
    # walk-order numbering: the left child follows its parent
    for i, n in enumerate(nodes):
        if n["count"] == 0:
            assert n["left"] == i + 1 and n["right"] == nodes[i + 1]["skipIndex"]


@pytest.mark.parametrize("name", list(SCENES))
def test_refit_without_moves_reproduces_the_uploaded_tree(orc, renderer, name):
    builder, cfg, w, h, spp = SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    host = s.arrays()
    st = renderer.update_instances([], [], T.REBUILD_FORCE_REFIT)
    assert st.action == T.REBUILD_FORCE_REFIT and st.tlas_nodes == len(host["tlasNodes"])
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == host["instances"].tobytes()
    assert idx.tolist() == host["tlasInstanceIndices"].tolist()
    assert _walk(nodes, idx) == _walk(host["tlasNodes"], host["tlasInstanceIndices"])
    so = orc.OrcScene(); builder(so)
    _check_frames(orc, renderer, so.desc(), cfg, w, h, spp)


@pytest.mark.parametrize("kind", ["rigid", "identity", "hostile"])
@pytest.mark.parametrize("name", list(SCENES))
def test_refit_after_moves(orc, renderer, name, kind):
    builder, cfg, w, h, spp = SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    so = orc.OrcScene(); builder(so)
    ids, xfs = _moves(len(so.arrays()["instances"]), kind)
    st = renderer.update_instances(ids, xfs, T.REBUILD_FORCE_REFIT)
    assert st.action == T.REBUILD_FORCE_REFIT
    for i, m in zip(ids, xfs):
        so.set_instance_transform(i, m)
        s.set_instance_transform(i, m)
    oa = so.arrays()
    nodes, idx, inst = _download(renderer)
    assert H.canon(inst).tobytes() == H.canon(oa["instances"]).tobytes(), "instance records derived on the device"
    assert H.canon(inst).tobytes() == H.canon(s.arrays()["instances"]).tobytes(), "... and by the library's host scene"
    want = _refit_numpy(oa["tlasNodes"], oa["tlasInstanceIndices"], oa["instances"])
    assert _walk(nodes, idx) == _walk(want, oa["tlasInstanceIndices"])
    general = any(not np.array_equal(np.frombuffer(bytes(m), np.float32), np.frombuffer(bytes(T.identity_affine()), np.float32)) for m in xfs) \
        or name == "rotated_mesh_and_sets"
    assert st.general_instances == (1 if general else 0)
    _check_frames(orc, renderer, _desc_with_tlas(so.desc(), nodes, idx, inst), cfg, w, h, spp)


@pytest.mark.parametrize("kind", ["rigid", "translate", "identity"])
@pytest.mark.parametrize("name", list(SCENES))
def test_rebuild_on_the_device(orc, renderer, name, kind):
    builder, cfg, w, h, spp = SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    so = orc.OrcScene(); builder(so)
    ids, xfs = _moves(len(so.arrays()["instances"]), kind)
    st = renderer.update_instances(ids, xfs, T.REBUILD_FORCE_REBUILD)
    assert st.action == T.REBUILD_FORCE_REBUILD and st.sah_cost > 0.0 and st.growth_final == 1.0 and st.growth_refit == 0.0
    for i, m in zip(ids, xfs):
        so.set_instance_transform(i, m)
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == so.arrays()["instances"].tobytes()
    _check_valid_tlas(nodes, idx, inst)
    assert st.tlas_nodes == len(nodes) and st.tlas_slots == len(idx)
    ref = _check_frames(orc, renderer, _desc_with_tlas(so.desc(), nodes, idx, inst), cfg, w, h, spp)
    # The same scene rebuilt on the host as Commit would (RebuildTLAS, Scene.cs:358-368): another tree, the same picture --
    # where the reference's picture does not depend on the tree.  It does for two of its quirks:
    # InvertRigidOrUniform writes the normalised columns of a rotation back as columns, i.e. not its inverse
    # (Scene.cs:628-631), so a rotated instance is seen somewhere else than its TLAS box; and TraceClosest prunes TLAS
    # boxes with closestT = tObj / uniformScale (SceneDeviceViews.cs:46-47,67), which for a scale > 1 lies before the box
    # the hit is in.  Translations and scales <= 1 are safe.
    if kind != "rigid" and name != "rotated_mesh_and_sets":
        so.rebuild_tlas()
        host, _ = _oracle_render(orc, so.desc(), cfg, w, h, spp)
        H.assert_outputs_equal(host, ref)
    # deterministic: the same moves again give the same tree
    renderer.update_instances(ids, xfs, T.REBUILD_FORCE_REBUILD)
    nodes2, idx2, inst2 = _download(renderer)
    assert nodes2.tobytes() == nodes.tobytes() and idx2.tolist() == idx.tolist() and inst2.tobytes() == inst.tobytes()
    # and a refit of the rebuilt tree keeps it
    st3 = renderer.update_instances([], [], T.REBUILD_FORCE_REFIT)
    nodes3, idx3, _ = _download(renderer)
    assert st3.action == T.REBUILD_FORCE_REFIT and nodes3.tobytes() == nodes.tobytes() and idx3.tolist() == idx.tolist()


def test_auto_refits_small_moves_and_rebuilds_when_the_tree_degrades(orc, renderer):
    builder, cfg, w, h, spp = SCENES["sphere_instances"]
    s = engine.Scene(); builder(s); renderer.commit(s)
    so = orc.OrcScene(); builder(so)
    n = len(so.arrays()["instances"])
    ids = list(range(1, n))
    small = [scenes.rotation_affine("y", 0.0, 1.0, (0.01 * (i % 3), 0.0, 0.0)) for i in ids]
    st = renderer.update_instances(ids, small, T.REBUILD_AUTO)
    assert st.action == T.REBUILD_FORCE_REBUILD and st.growth_final == 1.0, "an uploaded tree is replaced by the device-built one at the first Auto"
    st = renderer.update_instances(ids, [scenes.rotation_affine("y", 0.0, 1.0, (0.02 * (i % 3), 0.01, 0.0)) for i in ids], T.REBUILD_AUTO)
    assert st.action == T.REBUILD_FORCE_REFIT and 0.9 < st.growth_refit <= 1.5 and st.growth_final == st.growth_refit
    # every instance to the mirrored position: neighbours in the tree end up far apart
    far = [scenes.rotation_affine("y", 0.0, 1.0, (((i * 7919) % 13) - 6.0, 0.0, ((i * 104729) % 11) - 5.0)) for i in ids]
    st = renderer.update_instances(ids, far, T.REBUILD_AUTO)
    assert st.action == T.REBUILD_FORCE_REBUILD and st.growth_refit > 1.5 and st.growth_final == 1.0
    for i, m in zip(ids, far):
        so.set_instance_transform(i, m)
    nodes, idx, inst = _download(renderer)
    _check_valid_tlas(nodes, idx, inst)
    _check_frames(orc, renderer, _desc_with_tlas(so.desc(), nodes, idx, inst), cfg, w, h, spp)


def test_update_errors(renderer):
    fresh = engine.RTRenderer([0])
    try:
        with pytest.raises(engine.HrtError, match="no scene"):
            fresh.update_instances([], [], T.REBUILD_AUTO)
    finally:
        fresh.close()
    s = engine.Scene(); scenes.build_config2(s); renderer.commit(s)
    I = T.identity_affine()
    with pytest.raises(engine.HrtError, match="out of range"):
        renderer.update_instances([99], [I])
    with pytest.raises(engine.HrtError, match="twice"):
        renderer.update_instances([1, 1], [I, I])
    with pytest.raises(engine.HrtError, match="policy"):
        renderer.update_instances([1], [I], policy=7)
    cnt = (C.c_int64 * 3)()
    nodes = (T.BvhNode * 1)()
    with pytest.raises(engine.HrtError, match="too small"):
        renderer._check(engine.lib().hrt_scene_download_tlas(renderer._ctx, 0, nodes, 1, None, 0, None, 0, cnt))
    assert cnt[0] > 1


def _refit_numpy_fast(nodes, idx, inst):
    """_refit_numpy for big trees: walk-order numbering (children after parents), so one reverse sweep is bottom-up."""
    lo = np.stack([nodes["boundsMin"][f] for f in "XYZ"], axis=1).copy()
    hi = np.stack([nodes["boundsMax"][f] for f in "XYZ"], axis=1).copy()
    ilo = np.stack([inst["worldBoundsMin"][f] for f in "XYZ"], axis=1)
    ihi = np.stack([inst["worldBoundsMax"][f] for f in "XYZ"], axis=1)
    for i in range(len(nodes) - 1, -1, -1):
        n = nodes[i]
        if n["count"] > 0:
            ids = idx[n["first"]:n["first"] + n["count"]]
            lo[i], hi[i] = ilo[ids].min(axis=0), ihi[ids].max(axis=0)
        else:
            l, r = int(n["left"]), int(n["right"])
            assert l == i + 1 and r > l
            lo[i], hi[i] = np.minimum(lo[l], lo[r]), np.maximum(hi[l], hi[r])
    out = nodes.copy()
    for k, f in enumerate("XYZ"):
        out["boundsMin"][f] = lo[:, k]
        out["boundsMax"][f] = hi[:, k]
    return out


def test_big_tree_refit_and_rebuild(orc, renderer):
    """4001 instances: subtrees of more than 63 nodes take the arrival-counter climb of the refit kernel, the LBVH sorts
    real Morton keys (with duplicates: the jitter is quantised), the scan and cost kernels span many workgroups."""
    n = 4000
    s = engine.Scene(); scenes.build_random_spheres(s, n, extent=12.0); renderer.commit(s)
    so = orc.OrcScene(); scenes.build_random_spheres(so, n, extent=12.0)
    ids = np.arange(1, n + 1, dtype=np.int32)
    rng = np.random.default_rng(11)
    xf = np.zeros((n, 12), np.float32); xf[:, 0] = xf[:, 5] = xf[:, 10] = 1.0
    xf[:, [3, 7, 11]] = np.round(rng.uniform(-2.0, 2.0, (n, 3)) * 4.0).astype(np.float32) / 4.0
    for k in range(n):
        m = T.identity_affine(); m.m03, m.m13, m.m23 = float(xf[k, 3]), float(xf[k, 7]), float(xf[k, 11])
        so.set_instance_transform(int(ids[k]), m)
    oa = so.arrays()
    # refit: the uploaded topology (walk order) with recomputed boxes
    st = renderer.update_instances(ids, xf, T.REBUILD_FORCE_REFIT)
    nodes, idx, inst = _download(renderer)
    assert st.action == T.REBUILD_FORCE_REFIT and len(nodes) == len(oa["tlasNodes"]) > 63 * 32
    assert inst.tobytes() == oa["instances"].tobytes()
    assert nodes.tobytes() == _refit_numpy_fast(nodes, idx, inst).tobytes()
    want = _walk(_refit_numpy(oa["tlasNodes"], oa["tlasInstanceIndices"], oa["instances"]), oa["tlasInstanceIndices"])
    assert _walk(nodes, idx) == want
    # rebuild
    st = renderer.update_instances([], [], T.REBUILD_FORCE_REBUILD)
    nodes, idx, inst = _download(renderer)
    assert sorted(idx.tolist()) == list(range(n + 1)) and len(nodes) == st.tlas_nodes and len(nodes) % 2 == 1
    assert len(_walk(nodes, idx)) == len(nodes)
    assert nodes.tobytes() == _refit_numpy_fast(nodes, idx, inst).tobytes()
    cfg, w, h, spp = scenes.CONFIGS[3], 160, 96, 1
    ref, ost = _oracle_render(orc, _desc_with_tlas(so.desc(), nodes, idx, inst), cfg, w, h, spp)
    got, gst = _gpu_render(renderer, cfg, w, h, spp, T.FLAG_COUNTERS)
    H.assert_outputs_equal(ref, got)
    assert gst.k[1].as_dict() == ost.k[1].as_dict()
    so.rebuild_tlas()                                       # translations only: the host's tree shows the same picture
    host, _ = _oracle_render(orc, so.desc(), cfg, w, h, spp)
    H.assert_outputs_equal(host, ref)


# ------------------------------------------------------------------ deforming meshes: hrt_scene_update_positions
def _f3(v):
    return np.array(list(v.tolist()), np.float32)


def _refit_blas_numpy(arrs):
    """Boxes of every triangle-mesh BLAS recomputed for arrs['meshPositions'] (topology kept), then the world bounds of
    those instances (TransformAABB of the root box, Scene.cs:560-580, float32 in the reference's expression order)."""
    nodes, inst = arrs["blasNodes"].copy(), arrs["instances"].copy()
    pos = np.stack([arrs["meshPositions"][f] for f in "XYZ"], axis=1)
    tris = np.stack([arrs["meshTris"][f] for f in ("i0", "i1", "i2")], axis=1)
    prim = arrs["triPrimIdx"]

    def rec(i):
        n = nodes[i]
        if n["count"] > 0:
            v = pos[tris[prim[n["first"]:n["first"] + n["count"]]].reshape(-1)]
            lo, hi = np.minimum(v.min(axis=0), FMAX3), np.maximum(v.max(axis=0), -FMAX3)
        else:
            a, b = rec(int(n["left"])), rec(int(n["right"]))
            lo, hi = np.minimum(a[0], b[0]), np.maximum(a[1], b[1])
        for k, f in enumerate("XYZ"):
            nodes[i]["boundsMin"][f] = lo[k]
            nodes[i]["boundsMax"][f] = hi[k]
        return lo, hi

    import sys
    sys.setrecursionlimit(10000)
    for ii in range(len(inst)):
        if inst[ii]["type"] != 2 or inst[ii]["blasNodeCount"] <= 0:
            continue
        lo, hi = rec(int(inst[ii]["blasRoot"]))
        m = inst[ii]["objectToWorld"]
        rows = [[np.float32(m["m%d%d" % (r, c)]) for c in range(4)] for r in range(3)]
        corners = [(lo[0], lo[1], lo[2]), (hi[0], lo[1], lo[2]), (lo[0], hi[1], lo[2]), (lo[0], lo[1], hi[2]),
                   (hi[0], hi[1], lo[2]), (lo[0], hi[1], hi[2]), (hi[0], lo[1], hi[2]), (hi[0], hi[1], hi[2])]
        w = np.array([[((r[0] * c[0] + r[1] * c[1]) + r[2] * c[2]) + r[3] for r in rows] for c in corners], np.float32)
        for k, f in enumerate("XYZ"):
            inst[ii]["worldBoundsMin"][f] = min(w[:, k].min(), FMAX3[0]) if not np.isnan(w[:, k]).any() else np.float32("nan")
            inst[ii]["worldBoundsMax"][f] = max(w[:, k].max(), -FMAX3[0]) if not np.isnan(w[:, k]).any() else np.float32("nan")
    return nodes, inst


MESH_SCENES = {
    "blob_24x24": (lambda b: scenes.build_config4(b, 24, 24), scenes.CONFIGS[4], 128, 72, 2),
    "textured_alpha_meshes": (scenes.build_textured_test_scene, scenes.Config("t", 0, 0, 0, (0.3, 1.3, 4.2), (0.0, 0.7, 0.0)), 128, 80, 2),
    "rotated_mesh_and_sets": (scenes.build_rotated_instances_scene, CFG_ROT, 128, 80, 2),
}


@pytest.mark.parametrize("policy", [T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD])
@pytest.mark.parametrize("name", list(MESH_SCENES))
def test_deformed_meshes_are_refitted_on_the_device(orc, renderer, name, policy):
    builder, cfg, w, h, spp = MESH_SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    arrs = s.arrays()
    pos = np.stack([arrs["meshPositions"][f] for f in "XYZ"], axis=1)
    n = len(pos)
    first, cnt = n // 5, n - n // 5 - 3                                  # a sub-range: the rest keeps its place
    wob = (1.0 + 0.12 * np.sin(7.0 * pos[:, [1, 2, 0]] + 0.3)).astype(np.float32)
    new = pos.copy()
    new[first:first + cnt] = (pos * wob)[first:first + cnt]
    st = renderer.update_positions(first, new[first:first + cnt], policy)
    assert st.action == policy
    for k, f in enumerate("XYZ"):
        arrs["meshPositions"][f] = new[:, k]
    want_blas, want_inst = _refit_blas_numpy(arrs)
    assert renderer.download_array("meshPositions").tobytes() == arrs["meshPositions"].tobytes()
    assert renderer.download_array("blasNodes").tobytes() == want_blas.tobytes(), "BLAS boxes, uploaded numbering"
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == want_inst.tobytes(), "world bounds of the mesh instances"
    if policy == T.REBUILD_FORCE_REFIT:
        assert _walk(nodes, idx) == _walk(_refit_numpy(arrs["tlasNodes"], arrs["tlasInstanceIndices"], want_inst), arrs["tlasInstanceIndices"])
    else:
        _check_valid_tlas(nodes, idx, inst)
    arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = want_blas, inst, nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    _check_frames(orc, renderer, desc, cfg, w, h, spp)
    # nothing changes when the same positions come again, and n = 0 is allowed
    renderer.update_positions(first, new[first:first + cnt], T.REBUILD_FORCE_REFIT)
    renderer.update_positions(0, np.zeros((0, 3), np.float32), T.REBUILD_FORCE_REFIT)
    assert renderer.download_array("blasNodes").tobytes() == want_blas.tobytes()


def test_update_positions_errors(renderer):
    s = engine.Scene(); scenes.build_config4(s, 8, 8); renderer.commit(s)
    n = len(s.arrays()["meshPositions"])
    with pytest.raises(engine.HrtError, match="outside meshPositions"):
        renderer.update_positions(n - 1, np.zeros((2, 3), np.float32))
    with pytest.raises(engine.HrtError, match="outside meshPositions"):
        renderer.update_positions(-1, np.zeros((1, 3), np.float32))
    with pytest.raises(engine.HrtError, match="policy"):
        renderer.update_positions(0, np.zeros((1, 3), np.float32), policy=9)
    s2 = engine.Scene(); scenes.build_config2(s2); renderer.commit(s2)          # no meshes: only the empty range exists
    renderer.update_positions(0, np.zeros((0, 3), np.float32), T.REBUILD_FORCE_REFIT)
    with pytest.raises(engine.HrtError, match="outside meshPositions"):
        renderer.update_positions(0, np.zeros((1, 3), np.float32))


@pytest.mark.parametrize("name", list(MESH_SCENES))
def test_deformed_meshes_get_a_new_blas_on_the_device(orc, renderer, name):
    """policy | REBUILD_BLAS: every triangle-mesh BLAS is rebuilt (LBVH; subtrees of <= 4 triangles become leaves, more only
    if the tree would not fit otherwise) inside the node range and the leaf region it already owns.  The result is checked as a tree, its boxes against the numpy restatement, and frames against the
    oracle on the downloaded arrays."""
    builder, cfg, w, h, spp = MESH_SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    arrs = s.arrays()
    pos = np.stack([arrs["meshPositions"][f] for f in "XYZ"], axis=1)
    new = (pos * (1.0 + 0.1 * np.sin(6.0 * pos[:, [2, 0, 1]] + 0.5))).astype(np.float32)
    st = renderer.update_positions(0, new, T.REBUILD_FORCE_REFIT | T.REBUILD_BLAS)
    assert st.action == T.REBUILD_FORCE_REFIT
    got = {k: renderer.download_array(k) for k in ("meshPositions", "blasNodes", "triPrimIdx")}
    nodes, idx, inst = _download(renderer)
    for k, f in enumerate("XYZ"):
        arrs["meshPositions"][f] = new[:, k]
    assert got["meshPositions"].tobytes() == arrs["meshPositions"].tobytes()
    host_inst = arrs["instances"]
    for ii in range(len(inst)):
        a, b = inst[ii], host_inst[ii]
        if b["type"] != 2:
            assert a.tobytes() == b.tobytes()
            lo, hi = int(b["blasRoot"]), int(b["blasRoot"] + b["blasNodeCount"])
            assert got["blasNodes"][lo:hi].tobytes() == arrs["blasNodes"][lo:hi].tobytes(), "sphere BLASes are untouched"
            continue
        n_items = int(b["primIndexCount"])
        assert a["blasRoot"] == b["blasRoot"] and 0 < a["blasNodeCount"] <= b["blasNodeCount"] and a["blasNodeCount"] % 2 == 1
        items = sorted(arrs["triPrimIdx"][b["primIndexFirst"]:b["primIndexFirst"] + n_items].tolist())
        assert got["triPrimIdx"][b["primIndexFirst"]:b["primIndexFirst"] + n_items].tolist() == arrs["triPrimIdx"][b["primIndexFirst"]:b["primIndexFirst"] + n_items].tolist()
        # walk the new BLAS: every node once, leaves of <= 4, every triangle of the mesh in exactly one leaf
        root, cur, seen, tri, big = int(a["blasRoot"]), int(a["blasRoot"]), 0, [], 0
        while cur != -1:
            nd = got["blasNodes"][cur]
            seen += 1
            assert seen <= a["blasNodeCount"]
            if nd["count"] > 0:
                assert nd["count"] <= 14
                big = max(big, int(nd["count"]))
                tri.extend(got["triPrimIdx"][nd["first"]:nd["first"] + nd["count"]].tolist())
                cur = int(nd["skipIndex"])
            else:
                assert nd["left"] == cur + 1
                cur = int(nd["left"])
        assert seen == a["blasNodeCount"] and sorted(tri) == items
    arrs["blasNodes"], arrs["triPrimIdx"], arrs["instances"] = got["blasNodes"], got["triPrimIdx"], inst
    want_blas, want_inst = _refit_blas_numpy(arrs)
    assert want_blas.tobytes() == got["blasNodes"].tobytes() and want_inst.tobytes() == inst.tobytes()
    arrs["tlasNodes"], arrs["tlasInstanceIndices"] = nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    _check_frames(orc, renderer, desc, cfg, w, h, spp)
    # deterministic, and a refit afterwards keeps the new topology
    renderer.update_positions(0, new, T.REBUILD_FORCE_REFIT | T.REBUILD_BLAS)
    assert renderer.download_array("blasNodes").tobytes() == got["blasNodes"].tobytes()
    assert renderer.download_array("triPrimIdx").tobytes() == got["triPrimIdx"].tobytes()
    renderer.update_positions(0, new[:0], T.REBUILD_FORCE_REFIT)
    assert renderer.download_array("blasNodes").tobytes() == got["blasNodes"].tobytes()


def test_scene_manager_commit_honours_the_policy(orc, renderer):
    """SceneManager.Commit(policy) (SceneManager.cs:23 -> BvhManager.BuildOrRefit): full upload when the structure changed,
    in-place update of the device copy when only instances moved."""
    cfg, w, h, spp = scenes.Config("sm", 0, 0, 0, (0.0, 1.4, 4.5), (0.0, 0.5, 0.0)), 128, 80, 2
    m = engine.SceneManager(renderer)
    m.build_default_scene()
    m.commit()
    assert m.last_update is None
    so = orc.OrcScene(); so.build_default_scene()
    _check_frames(orc, renderer, so.desc(), cfg, w, h, spp)
    move = scenes.rotation_affine("y", 0.0, 1.0, (0.4, 0.3, -0.6))
    m.set_instance_transform(2, move)
    m.commit(T.REBUILD_FORCE_REFIT)
    assert m.last_update is not None and m.last_update.action == T.REBUILD_FORCE_REFIT and m.last_update.general_instances == 1
    so.set_instance_transform(2, move)
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == so.arrays()["instances"].tobytes() == m.scene.arrays()["instances"].tobytes()
    _check_frames(orc, renderer, _desc_with_tlas(so.desc(), nodes, idx, inst), cfg, w, h, spp)
    m.commit(T.REBUILD_FORCE_REBUILD)                       # nothing moved, nothing added: nothing to do
    assert renderer.download_tlas()[3][0] == len(nodes)
    # a new instance changes the structure: everything is rebuilt on the host and uploaded, as the reference always does
    sid = m.scene.add_sphere(scenes.sphere((0.3, 1.6, 0.2), 0.3, (0.9, 0.6, 0.1)))
    m.scene.build_sphere_instance([sid])
    m.commit()
    assert m.last_update is None
    sid2 = so.add_sphere(scenes.sphere((0.3, 1.6, 0.2), 0.3, (0.9, 0.6, 0.1)))
    so.build_sphere_instance([sid2]); so.rebuild_tlas()
    _check_frames(orc, renderer, so.desc(), cfg, w, h, spp)
    with pytest.raises(ValueError):
        engine.SceneManager(None)


def test_refit_of_a_tree_in_builder_numbering(orc, renderer, monkeypatch):
    if b"tuning-build" not in engine.lib().hrt_version():
        pytest.skip("HRT_BUILDER_ORDER is an A/B knob of tuning builds (make variant NAME=tuning DEFS=-DHRT_TUNING); the shipped library always renumbers into walk order")
    _refit_builder_numbering(orc, renderer, monkeypatch)


def _refit_builder_numbering(orc, renderer, monkeypatch):
    """HRT_BUILDER_ORDER keeps the uploaded numbering, where a subtree is no index range: the refit then climbs with arrival
    counters (release / acquire at agent scope) from every leaf.  2001 instances: 2357 nodes, eleven levels."""
    monkeypatch.setenv("HRT_BUILDER_ORDER", "1")
    n = 2000
    s = engine.Scene(); scenes.build_random_spheres(s, n, extent=9.0); renderer.commit(s)
    so = orc.OrcScene(); scenes.build_random_spheres(so, n, extent=9.0)
    ids = np.arange(1, n + 1, dtype=np.int32)
    rng = np.random.default_rng(5)
    xf = np.zeros((n, 12), np.float32); xf[:, 0] = xf[:, 5] = xf[:, 10] = 1.0
    xf[:, [3, 7, 11]] = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    for k in range(n):
        m = T.identity_affine(); m.m03, m.m13, m.m23 = float(xf[k, 3]), float(xf[k, 7]), float(xf[k, 11])
        so.set_instance_transform(int(ids[k]), m)
    st = renderer.update_instances(ids, xf, T.REBUILD_FORCE_REFIT)
    assert st.action == T.REBUILD_FORCE_REFIT
    oa = so.arrays()
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == oa["instances"].tobytes()
    import sys
    sys.setrecursionlimit(10000)
    assert _walk(nodes, idx) == _walk(_refit_numpy(oa["tlasNodes"], oa["tlasInstanceIndices"], oa["instances"]), oa["tlasInstanceIndices"])
    monkeypatch.delenv("HRT_BUILDER_ORDER")
    s2 = engine.Scene(); scenes.build_config2(s2); renderer.commit(s2)          # leave the shared renderer in its default state


# ------------------------------------------------------------------ moving spheres: hrt_scene_update_spheres
def _refit_sphere_blas_numpy(arrs):
    """Boxes of every sphere-set BLAS recomputed for arrs['spheres'] (topology kept: the union of centre -+ radius over the
    spheres each node really holds), then the world bounds of those instances."""
    nodes, inst = arrs["blasNodes"].copy(), arrs["instances"].copy()
    sp, prim = arrs["spheres"], arrs["spherePrimIdx"]

    def rec(i):
        n = nodes[i]
        if n["count"] > 0:
            los, his = [], []
            for sid in prim[n["first"]:n["first"] + n["count"]]:
                c, r = _f3(sp[sid]["center"]), np.float32(sp[sid]["radius"])
                los.append(c - r); his.append(c + r)
            lo, hi = np.min(np.stack([FMAX3] + los), axis=0), np.max(np.stack([-FMAX3] + his), axis=0)
        else:
            a, b = rec(int(n["left"])), rec(int(n["right"]))
            lo, hi = np.minimum(a[0], b[0]), np.maximum(a[1], b[1])
        for k, f in enumerate("XYZ"):
            nodes[i]["boundsMin"][f] = lo[k]
            nodes[i]["boundsMax"][f] = hi[k]
        return lo, hi

    for ii in range(len(inst)):
        if inst[ii]["type"] != 1 or inst[ii]["blasNodeCount"] <= 0:
            continue
        lo, hi = rec(int(inst[ii]["blasRoot"]))
        m = inst[ii]["objectToWorld"]
        rows = [[np.float32(m["m%d%d" % (r, c)]) for c in range(4)] for r in range(3)]
        corners = [(lo[0], lo[1], lo[2]), (hi[0], lo[1], lo[2]), (lo[0], hi[1], lo[2]), (lo[0], lo[1], hi[2]),
                   (hi[0], hi[1], lo[2]), (lo[0], hi[1], hi[2]), (hi[0], lo[1], hi[2]), (hi[0], hi[1], hi[2])]
        w = np.array([[((r[0] * c[0] + r[1] * c[1]) + r[2] * c[2]) + r[3] for r in rows] for c in corners], np.float32)
        for k, f in enumerate("XYZ"):
            inst[ii]["worldBoundsMin"][f] = min(w[:, k].min(), FMAX3[0]) if not np.isnan(w[:, k]).any() else np.float32("nan")
            inst[ii]["worldBoundsMax"][f] = max(w[:, k].max(), -FMAX3[0]) if not np.isnan(w[:, k]).any() else np.float32("nan")
    return nodes, inst


def _nine_in_one(b):
    ids = [b.add_sphere(scenes.sphere((x, 0.3 * (i % 3), -0.4 * i), 0.45, (0.3 + 0.07 * i, 0.9 - 0.08 * i, 0.5)))
           for i, x in enumerate([3.0, -2.0, 0.5, -4.0, 2.0, 1.0, -1.0, 4.0, -3.0])]
    g = b.add_sphere(scenes.sphere((0.0, -200.0, 0.0), 199.5, (0.7, 0.7, 0.7)))
    b.build_sphere_instance(ids)
    b.build_sphere_instance([g])
    b.rebuild_tlas()


SPHERE_SCENES = {
    "sphere_instances": SCENES["sphere_instances"],
    "cornell_flat_leaves": SCENES["cornell_flat_leaves"],
    "nine_spheres_in_one_blas": (_nine_in_one, scenes.Config("q", 0, 0, 0, (0.0, 1.5, 7.0), (0.0, 0.2, -1.5)), 128, 80, 2),
    "rotated_mesh_and_sets": SCENES["rotated_mesh_and_sets"],
}


@pytest.mark.parametrize("policy", [T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD])
@pytest.mark.parametrize("name", list(SPHERE_SCENES))
def test_moved_spheres_are_refitted_on_the_device(orc, renderer, name, policy):
    builder, cfg, w, h, spp = SPHERE_SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    arrs = s.arrays()
    sp = arrs["spheres"].copy()
    n = len(sp)
    first, cnt = 1, n - 1                                             # sphere 0 (a ground / wall) stays
    k = np.arange(n, dtype=np.float32)
    sp["center"]["X"][first:] += (0.25 * np.sin(1.7 * k))[first:]
    sp["center"]["Y"][first:] += (0.15 * np.cos(0.9 * k) + 0.1)[first:]
    sp["center"]["Z"][first:] -= (0.2 * np.sin(0.4 * k + 1.0))[first:]
    small = sp["radius"] < 10.0
    small[:first] = False
    sp["radius"][small] *= (0.8 + 0.05 * (k % 7))[small].astype(np.float32)
    sp["albedo"]["X"][first:] = (0.2 + 0.05 * (k % 13))[first:]
    st = renderer.update_spheres(first, sp[first:first + cnt], policy)
    assert st.action == policy
    arrs["spheres"] = sp
    want_blas, want_inst = _refit_sphere_blas_numpy(arrs)
    assert renderer.download_array("spheres").tobytes() == sp.tobytes()
    assert renderer.download_array("blasNodes").tobytes() == want_blas.tobytes()
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == want_inst.tobytes()
    if name in ("sphere_instances", "cornell_flat_leaves"):
        assert st.general_instances == 0, "single spheres under an identity transform stay on the fast path"
    if policy == T.REBUILD_FORCE_REFIT:
        assert _walk(nodes, idx) == _walk(_refit_numpy(arrs["tlasNodes"], arrs["tlasInstanceIndices"], want_inst), arrs["tlasInstanceIndices"])
    else:
        _check_valid_tlas(nodes, idx, inst)
    arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = want_blas, inst, nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    ref = _check_frames(orc, renderer, desc, cfg, w, h, spp)
    if name in ("sphere_instances", "cornell_flat_leaves"):
        # single-sphere instances: the reference's builder makes the same boxes, so a scene built from the moved spheres on
        # the host shows the same picture (another TLAS, no rotations, no scales)
        so = orc.OrcScene()

        class Moved:
            def __init__(self, inner): self.inner, self.k = inner, 0
            def add_sphere(self, sph):
                q = T.Sphere.from_buffer_copy(sp[self.k].tobytes()); self.k += 1
                return self.inner.add_sphere(q)
            def __getattr__(self, a): return getattr(self.inner, a)
        builder(Moved(so))
        assert so.arrays()["spheres"].tobytes() == sp.tobytes()
        host, _ = _oracle_render(orc, so.desc(), cfg, w, h, spp)
        H.assert_outputs_equal(host, ref)


@pytest.mark.parametrize("name", ["sphere_instances", "cornell_flat_leaves", "nine_spheres_in_one_blas"])
def test_non_finite_spheres_through_the_update_path(orc, renderer, name):
    """NaN / infinite centres and radii written AFTER the upload: the unions the device recomputes are the host's Math.Min / Max
    (a NaN operand is returned), which can leave a node box that its own children do not lie inside as far as the slab test is
    concerned -- walk organisations that skip inner-node tests (leaf sweep, second tree) have to stand down for such a tree."""
    builder, cfg, w, h, spp = SPHERE_SCENES[name]
    s = engine.Scene(); builder(s); renderer.commit(s)
    arrs = s.arrays()
    sp = arrs["spheres"].copy()
    nan, inf = np.float32("nan"), np.float32("inf")
    edits = [("X", nan, None), ("Y", inf, None), (None, None, inf), ("X", inf, inf), (None, None, nan), (None, None, np.float32(-0.3)), ("Z", -inf, None)]
    for k, (axis, cv, rv) in enumerate(edits):
        i = 1 + k
        if i >= len(sp):
            break
        if axis is not None: sp["center"][axis][i] = cv
        if rv is not None: sp["radius"][i] = rv
    st = renderer.update_spheres(1, sp[1:], T.REBUILD_FORCE_REFIT)
    assert st.action == T.REBUILD_FORCE_REFIT
    arrs["spheres"] = sp
    with np.errstate(all="ignore"):
        want_blas, want_inst = _refit_sphere_blas_numpy(arrs)
        want_tlas = _refit_numpy(arrs["tlasNodes"], arrs["tlasInstanceIndices"], want_inst)
    assert H.canon(renderer.download_array("blasNodes")).tobytes() == H.canon(want_blas).tobytes()
    nodes, idx, inst = _download(renderer)
    assert H.canon(inst).tobytes() == H.canon(want_inst).tobytes()
    assert _walk(nodes, idx) == _walk(want_tlas, arrs["tlasInstanceIndices"])
    arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = want_blas, inst, nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    _check_frames(orc, renderer, desc, cfg, w, h, spp)
    s2 = engine.Scene(); scenes.build_config2(s2); renderer.commit(s2)          # leave the shared renderer in its default state


@pytest.mark.parametrize("slots", [1, 3])
def test_second_tree_stands_down_and_returns(orc, renderer, slots):
    """400 one-sphere instances (a second tree exists): an instance gets a translation (a general instance: the second tree cannot
    describe the scene), later the identity again (it can), with sphere moves in between -- every frame against the oracle over the
    tree the device holds, in the organisations that use the second tree and one that does not.  slots = 3: one context over three
    device slots on the one GPU (every slot keeps and refits a second tree of its own)."""
    if slots > 1:
        renderer = engine.RTRenderer([0] * slots)
    cfg = scenes.Config("st", 0, 0, 0, (0.0, 2.2, 7.5), (0.0, 0.6, 0.0))
    w, h, spp = 96, 60, 2
    s = engine.Scene(); scenes.build_random_spheres(s, 400, seed=0xABCD, extent=3.0); renderer.commit(s)
    so = orc.OrcScene(); scenes.build_random_spheres(so, 400, seed=0xABCD, extent=3.0)
    arrs = so.arrays()

    def check():
        nodes, idx, inst = _download(renderer)
        arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = renderer.download_array("blasNodes"), inst, nodes, idx
        desc, keep = T.scene_desc_from_arrays(arrs)
        ref, ost = _oracle_render(orc, desc, cfg, w, h, spp)
        for flags in (0, T.FLAG_STREAMED, T.FLAG_STREAMED | T.FLAG_COUNTERS):
            got, gst = _gpu_render(renderer, cfg, w, h, spp, flags)
            H.assert_outputs_equal(ref, got)

    check()
    moved = scenes.rotation_affine("y", 0.0, 1.0, (0.3, 0.1, -0.2))
    st = renderer.update_instances([7], [moved], T.REBUILD_FORCE_REFIT)
    assert st.general_instances == 1
    check()
    sp = arrs["spheres"].copy()
    sp["center"]["X"][1:] += np.float32(0.05)
    renderer.update_spheres(1, sp[1:], T.REBUILD_AUTO); arrs["spheres"] = sp
    check()
    st = renderer.update_instances([7], [T.identity_affine()], T.REBUILD_FORCE_REFIT)
    assert st.general_instances == 0
    check()
    sp = sp.copy(); sp["radius"][1:] *= np.float32(0.9)
    renderer.update_spheres(1, sp[1:], T.REBUILD_FORCE_REBUILD); arrs["spheres"] = sp
    check()
    s2 = engine.Scene(); scenes.build_config2(s2); renderer.commit(s2)          # leave the shared renderer in its default state


def test_update_spheres_errors(renderer):
    s = engine.Scene(); scenes.build_config2(s); renderer.commit(s)
    n = len(s.arrays()["spheres"])
    one = s.arrays()["spheres"][:1]
    with pytest.raises(engine.HrtError, match="outside spheres"):
        renderer.update_spheres(n, one)
    with pytest.raises(engine.HrtError, match="outside spheres"):
        renderer.update_spheres(-1, one)
    with pytest.raises(engine.HrtError, match="policy"):
        renderer.update_spheres(0, one, policy=5)
    renderer.update_spheres(0, one[:0], T.REBUILD_FORCE_REFIT)


def test_auto_rebuilds_a_mesh_blas_when_its_boxes_grow(orc, renderer):
    builder, cfg, w, h, spp = MESH_SCENES["blob_24x24"]
    s = engine.Scene(); builder(s); renderer.commit(s)
    arrs = s.arrays()
    pos = np.stack([arrs["meshPositions"][f] for f in "XYZ"], axis=1)
    host_nodes = int(arrs["instances"][arrs["instances"]["type"] == 2][0]["blasNodeCount"])
    st = renderer.update_positions(0, (pos * np.float32(1.01)).astype(np.float32), T.REBUILD_AUTO)
    assert st.blas_action == T.REBUILD_FORCE_REFIT and 0.9 < st.blas_growth < 1.5
    assert renderer.download_array("triPrimIdx").tobytes() == arrs["triPrimIdx"].tobytes()
    # every vertex to a pseudo-random place inside the old bounds: neighbours in the tree end up far apart
    rng = np.random.default_rng(3)
    new = (pos[rng.permutation(len(pos))]).astype(np.float32)
    st = renderer.update_positions(0, new, T.REBUILD_AUTO)
    assert st.blas_action == T.REBUILD_FORCE_REBUILD and st.blas_growth > 1.5
    got = {k: renderer.download_array(k) for k in ("blasNodes", "triPrimIdx")}
    nodes, idx, inst = _download(renderer)
    mi = inst[inst["type"] == 2][0]
    assert 0 < mi["blasNodeCount"] <= host_nodes
    for k, f in enumerate("XYZ"):
        arrs["meshPositions"][f] = new[:, k]
    arrs["blasNodes"], arrs["triPrimIdx"], arrs["instances"] = got["blasNodes"], got["triPrimIdx"], inst
    want_blas, want_inst = _refit_blas_numpy(arrs)
    assert want_blas.tobytes() == got["blasNodes"].tobytes() and want_inst.tobytes() == inst.tobytes()
    arrs["tlasNodes"], arrs["tlasInstanceIndices"] = nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    _check_frames(orc, renderer, desc, cfg, w, h, spp)
    # the rebuilt tree is the new base: the same positions again only refit
    st = renderer.update_positions(0, new, T.REBUILD_AUTO)
    assert st.blas_action == T.REBUILD_FORCE_REFIT and abs(st.blas_growth - 1.0) < 1e-3


def test_big_mesh_refit_and_rebuild(orc, renderer):
    """Config 4's 100 352-triangle mesh (65 535 BLAS nodes): the level launches of the refit above the direct subtrees, the
    radix sort of real Morton keys and the leaf-limit search of the BLAS rebuild, checked against the numpy restatement and
    the oracle at a small resolution."""
    builder, cfg, w, h, spp = (lambda b: scenes.build_config4(b)), scenes.CONFIGS[4], 160, 90, 1
    s = engine.Scene(); builder(s); renderer.commit(s)
    arrs = s.arrays()
    pos = np.stack([arrs["meshPositions"][f] for f in "XYZ"], axis=1)
    new = (pos * (1.0 + 0.05 * np.sin(4.0 * pos[:, [1, 2, 0]] + 0.2))).astype(np.float32)
    st = renderer.update_positions(0, new, T.REBUILD_FORCE_REFIT)
    assert st.blas_action == T.REBUILD_FORCE_REFIT
    for k, f in enumerate("XYZ"):
        arrs["meshPositions"][f] = new[:, k]
    want_blas, want_inst = _refit_blas_numpy(arrs)
    assert renderer.download_array("blasNodes").tobytes() == want_blas.tobytes()
    nodes, idx, inst = _download(renderer)
    assert inst.tobytes() == want_inst.tobytes()
    arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = want_blas, inst, nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    ref, ost = _oracle_render(orc, desc, cfg, w, h, spp)
    got, gst = _gpu_render(renderer, cfg, w, h, spp, T.FLAG_COUNTERS)
    H.assert_outputs_equal(ref, got)
    assert gst.k[1].as_dict() == ost.k[1].as_dict()
    # new topology on the device
    st = renderer.update_positions(0, new[:0], T.REBUILD_FORCE_REFIT | T.REBUILD_BLAS)
    got_b = {k: renderer.download_array(k) for k in ("blasNodes", "triPrimIdx")}
    nodes, idx, inst = _download(renderer)
    mi = np.nonzero(inst["type"] == 2)[0][0]
    n_items = int(inst[mi]["primIndexCount"])
    assert 0 < inst[mi]["blasNodeCount"] <= arrs["instances"][mi]["blasNodeCount"]
    region = got_b["triPrimIdx"][n_items:2 * n_items] if int(inst[mi]["primIndexFirst"]) == 0 else None
    arrs["blasNodes"], arrs["triPrimIdx"], arrs["instances"] = got_b["blasNodes"], got_b["triPrimIdx"], inst
    want_blas, want_inst = _refit_blas_numpy(arrs)
    assert want_blas.tobytes() == got_b["blasNodes"].tobytes() and want_inst.tobytes() == inst.tobytes()
    if region is not None:
        assert sorted(region.tolist()) == sorted(arrs["triPrimIdx"][:n_items].tolist()), "every triangle in exactly one leaf slot"
    arrs["tlasNodes"], arrs["tlasInstanceIndices"] = nodes, idx
    desc, keep = T.scene_desc_from_arrays(arrs)
    ref, ost = _oracle_render(orc, desc, cfg, w, h, spp)
    for flags in (T.FLAG_COUNTERS, 0, T.FLAG_COUNTERS | T.FLAG_STREAMED | T.FLAG_REFERENCE_LAYOUT):
        got, gst = _gpu_render(renderer, cfg, w, h, spp, flags)
        H.assert_outputs_equal(ref, got)
        if flags & T.FLAG_COUNTERS:
            assert gst.k[1].as_dict() == ost.k[1].as_dict()
