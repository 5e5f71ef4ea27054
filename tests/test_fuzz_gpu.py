"""Seeded differential fuzz of the render path: random small scenes (sphere sets with every shading kind, multi-sphere instances built
before and after their spheres' neighbours, rigid / scaled / identity transforms, 0-3 textured or alpha-cut meshes), random cameras,
lights and frame parameters (spp, maxDepth 0..7, frame number, locked noise, ReSTIR reuse over two frames) -- every output array of
every frame against the oracle, in the kernel organisation the library picks and in one forced organisation per case.
HRT_FUZZ_CASES / HRT_FUZZ_SEED override the number of cases and the first seed (one-off deep runs of this round: 3 000 + 20 000 cases)."""
import os

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, scenes, engine
from tests import helpers as H

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("HRT_FUZZ_CASES", "0")) or 120
SEED0 = int(os.environ.get("HRT_FUZZ_SEED", "0"), 0) or 0x5EED0000
FORCED = [T.FLAG_MEGAKERNEL, T.FLAG_STREAMED, T.FLAG_STREAMED | T.FLAG_COUNTERS, T.FLAG_MEGAKERNEL | T.FLAG_REFERENCE_LAYOUT, T.FLAG_STREAMED | T.FLAG_REFERENCE_LAYOUT]


def _scene_recipe(seed):
    """A list of builder operations (applied alike to the oracle's scene and the library's) + frame settings."""
    rng = np.random.default_rng(seed)
    ops = []
    n_sph = int(rng.integers(1, 28))
    shading = lambda: int(rng.choice([T.SHADING_LAMBERT] * 5 + [T.SHADING_MIRROR, T.SHADING_GLASS]))
    if rng.random() < 0.8:
        ops.append(("sphere", (0.0, -200.0, 0.0), 200.0, tuple(float(v) for v in rng.uniform(0.3, 0.9, 3)), T.SHADING_LAMBERT, 1.0, True))
    pending = []
    for i in range(n_sph):
        c = (float(rng.uniform(-3, 3)), float(rng.uniform(0.05, 2.0)), float(rng.uniform(-3, 3)))
        ops.append(("sphere", c, float(rng.uniform(0.08, 0.7)), tuple(float(v) for v in rng.uniform(0.05, 1.0, 3)), shading(), float(rng.choice([1.0, 1.33, 1.5, 2.4])), False))
        pending.append(i)
        if rng.random() < 0.45 or i == n_sph - 1:          # flush the pending spheres as one or several instances
            while pending:
                k = int(rng.integers(1, min(len(pending), 6) + 1))
                group, pending = pending[:k], pending[k:]
                kind = rng.random()
                if kind < 0.6: xf = None
                elif kind < 0.8: xf = ("y", 0.0, 1.0, tuple(float(v) for v in rng.uniform(-0.5, 0.5, 3)))
                else: xf = (str(rng.choice(list("xyz"))), float(rng.uniform(-80, 80)), float(rng.choice([0.5, 0.8, 1.0, 1.3])), tuple(float(v) for v in rng.uniform(-0.5, 0.5, 3)))
                ops.append(("instance", group, xf))
    for m in range(int(rng.choice([0, 0, 1, 1, 2, 3]))):
        n = int(rng.integers(1, 6))
        s = np.linspace(-1.0, 1.0, n + 1)
        yq, xq = np.meshgrid(s * rng.uniform(0.3, 1.0) + rng.uniform(0.3, 1.5), s * rng.uniform(0.3, 1.2) + rng.uniform(-1.5, 1.5), indexing="ij")
        zq = rng.uniform(-2.0, 1.0) + rng.uniform(0.0, 0.3) * np.sin(rng.uniform(1, 4) * xq) * np.cos(rng.uniform(1, 4) * yq)
        tex = rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4), dtype=np.uint8)
        mask = rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4), dtype=np.uint8)
        mat = dict(kd=tuple(float(v) for v in rng.uniform(0.1, 1.0, 3)), diffuse_tex=int(rng.choice([-1, 0])), alpha_tex=int(rng.choice([-1, -1, 1])),
                   two_sided=int(rng.integers(0, 2)), alpha_cutoff=float(rng.choice([0.3, 0.5, 0.7])))
        xf = None if rng.random() < 0.5 else (str(rng.choice(list("xyz"))), float(rng.uniform(-60, 60)), float(rng.choice([0.6, 1.0, 1.4])), tuple(float(v) for v in rng.uniform(-0.6, 0.6, 3)))
        uvs = (float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 3.0)))
        ops.append(("mesh", xq, yq, zq, uvs, mat, tex, mask, xf))
    frame = dict(origin=tuple(float(v) for v in (rng.uniform(-1.5, 1.5), rng.uniform(0.4, 3.0), rng.uniform(3.0, 7.0))),
                 lookat=tuple(float(v) for v in (rng.uniform(-0.5, 0.5), rng.uniform(0.2, 1.2), rng.uniform(-0.5, 0.5))),
                 vfov=float(rng.choice([35.0, 60.0, 90.0])), max_depth=int(rng.choice([0, 1, 2, 3, 3, 3, 4, 5, 7])), spp=int(rng.integers(1, 4)),
                 frame=int(rng.choice([0, 1, 7, 123456, -3])), lock=int(rng.choice([0, 0, 0, 987654321, -5])), reuse=bool(rng.random() < 0.35),
                 sun=(float(rng.uniform(0, 6.28)), float(rng.uniform(0.1, 1.5))), w=int(rng.choice([32, 40, 56])), h=int(rng.choice([24, 32, 40])))
    return ops, frame


NAN, INF = float("nan"), float("inf")


def _poison(ops, fr, seed):
    """The recipe again with non-finite, zero, negative and huge numbers planted where nobody would author them: sphere centres,
    radii, albedos and IORs, instance scales, angles and translations, single mesh vertices, alpha cutoffs, the camera's field of
    view, the sun's elevation.  IEEE arithmetic gives every one of them SOME result in the reference; the bits have to be the same.
    One NaN or infinity in a box reaches the TLAS root through the host's Min / Max and hides the whole scene (every slab test
    fails), so two scenes in three get their GEOMETRY poisoned with finite numbers only (zero, negative, huge, tiny), and the rest
    with anything; what does not feed a box (albedo, IOR, shading id, alpha cutoff) is poisoned with anything in every scene."""
    rng = np.random.default_rng(seed ^ 0xBAD5EED)
    pick = lambda pool: float(pool[int(rng.integers(0, len(pool)))])
    wild = rng.random() < 0.33
    coord = [NAN, INF, -INF, 1e30, -1e30] if wild else [1e30, -1e30, 1e18, 0.0, 3e-39]
    radius = lambda r: pick([0.0, -r, INF, NAN, 1e-30, 1e30, -INF] if wild else [0.0, -r, -3.0 * r, 1e-30, 3e-39, 1e30, 50.0])
    scale = [0.0, -1.0, -0.5, 1e19, 1e-19, NAN, INF] if wild else [0.0, -1.0, -0.5, 1e19, 1e-19, 3.0]
    one_of3 = lambda v, pool: tuple(pick(pool) if i == int(rng.integers(0, 3)) else x for i, x in enumerate(v))
    def xf_poison(xf):
        if xf is None or rng.random() >= 0.2: return xf
        axis, ang, sc, tr = xf
        k = int(rng.integers(0, 3))
        if k == 0: sc = pick(scale)
        elif k == 1: ang = pick([1e9, -720.0, 1e30, 90.0])
        else: tr = one_of3(tr, coord)
        return (axis, ang, sc, tr)
    out = []
    for op in ops:
        if op[0] == "sphere":
            _, c, r, kd, sh, ior, own = op
            if rng.random() < 0.07: c = one_of3(c, coord)
            if rng.random() < 0.08: r = radius(r)
            if rng.random() < 0.1: kd = one_of3(kd, [NAN, INF, -1.0, 0.0, 1e30])
            if rng.random() < 0.1: ior = pick([NAN, 0.0, INF, -1.5, 1.0, 1e-30, 40.0])
            if rng.random() < 0.05: sh = int(rng.choice([-1, 3, 7, 2 ** 31 - 1]))
            out.append(("sphere", c, r, kd, sh, ior, own))
        elif op[0] == "instance":
            out.append(("instance", op[1], xf_poison(op[2])))
        else:
            _, xq, yq, zq, uvs, mat, tex, mask, xf = op
            if rng.random() < 0.3:
                zq = zq.copy()
                zq.flat[int(rng.integers(0, zq.size))] = pick(coord)
            if rng.random() < 0.15:
                xq = xq.copy()
                xq.flat[int(rng.integers(0, xq.size))] = pick(coord)                    # texture coordinates derive from xq too
            if rng.random() < 0.3:
                mat = dict(mat); mat["alpha_cutoff"] = pick([NAN, 2.0, -1.0, INF, 0.0, 1.0])
            out.append(("mesh", xq, yq, zq, uvs, mat, tex, mask, xf_poison(xf)))
    fr = dict(fr)
    if rng.random() < 0.08: fr["vfov"] = pick([0.0, 179.9999, 180.0, 360.0, -60.0])
    if rng.random() < 0.08: fr["sun"] = (fr["sun"][0], pick([INF, NAN, 0.0, -1.0, 1.5707964]))
    if rng.random() < 0.04: fr["origin"] = tuple(pick([1e30, INF]) if i == 1 else v for i, v in enumerate(fr["origin"]))
    return out, fr


def _apply(b, ops):
    ids = []
    for op in ops:
        if op[0] == "sphere":
            _, c, r, kd, sh, ior, own = op
            sid = b.add_sphere(scenes.sphere(c, r, kd, sh, ior))
            if own: b.build_sphere_instance([sid])
            else: ids.append(sid)
        elif op[0] == "instance":
            _, group, xf = op
            b.build_sphere_instance([ids[g] for g in group], None if xf is None else scenes.rotation_affine(*xf))
        else:
            _, xq, yq, zq, uvs, mat, tex, mask, xf = op
            g = scenes.grid_mesh(xq, yq, zq, (xq - xq.min()) * uvs[0], (yq - yq.min()) * uvs[1])
            m = scenes.material(**mat)
            b.load_mesh_instance(scenes.MeshData(g.positions, g.triangles, g.texcoords, g.tri_uvs, [m], None, [tex, mask]), None if xf is None else scenes.rotation_affine(*xf))
    b.rebuild_tlas()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("hostile", [False, True])
def test_random_scenes_match_the_oracle(orc, renderer, hostile):
    failures = []
    for case in range(N_CASES if not hostile else max(8, N_CASES // 2)):
        seed = SEED0 + case + (0x700000 if hostile else 0)
        ops, fr = _scene_recipe(seed)
        if hostile:
            ops, fr = _poison(ops, fr, seed)
        so = orc.OrcScene(); _apply(so, ops)
        s = engine.Scene(); _apply(s, ops)
        for k_, a_ in so.arrays().items():
            assert H.canon(a_).tobytes() == H.canon(s.arrays()[k_]).tobytes(), (seed, k_)
        renderer.commit(s)
        cfg = scenes.Config("fz", fr["w"], fr["h"], fr["spp"], fr["origin"], fr["lookat"], max_depth=fr["max_depth"], vfov=fr["vfov"],
                            extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
        w, h = fr["w"], fr["h"]
        for fl in (0, FORCED[case % len(FORCED)]):
            renderer.reset_history()
            # every third streamed case runs in sample batches: a workspace of one, or one and a half, samples' paths
            batches = (fl & T.FLAG_STREAMED) and fr["spp"] > 1 and case % 3 == 0
            renderer.set_workspace_limit((w * h * (2 if case % 2 else 3)) // 2 + 7 if batches else 0)
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
                ost = orc.render_frame(so.desc(), po_, oo, po)
                got, og = T.alloc_outputs(w, h)
                st = renderer.render_params(pg_, og, flags=fl)
                bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
                bad = {k: v for k, v in bad.items() if v}
                if not bad and (fl & T.FLAG_COUNTERS) and st.k[1].as_dict() != ost.k[1].as_dict():
                    bad = {"counters": 1}
                if bad:
                    failures.append((seed, fl, f, bad))
                    break
        renderer.set_workspace_limit(0)
        if len(failures) >= 5:
            break
    assert not failures, "cases that differ from the oracle (seed, flags, frame, {array: elements}): %s" % failures


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("hostile", [False, True])
def test_random_many_sphere_scenes_on_the_second_tree(orc, renderer, hostile):
    """256..700 one-sphere instances (the second, device-built TLAS is in use for every production walk), some of them exact
    duplicates or near-duplicates of others (equal and almost equal distances), random extents and cameras."""
    n_cases = max(6, N_CASES // 12)
    failures = []
    for case in range(n_cases):
        rng = np.random.default_rng(SEED0 + 0x100000 + case + (0x40000 if hostile else 0))
        n = int(rng.integers(256, 700))
        ext = float(rng.choice([2.0, 5.0, 12.0]))
        recs = [((0.0, -500.0, 0.0), 500.0, (0.6, 0.6, 0.6), T.SHADING_LAMBERT, 1.0)] if rng.random() < 0.7 else []
        for i in range(n):
            if recs and i > 4 and rng.random() < 0.15:           # a twin: same place, or a hair away, another material
                c, r, _, _, _ = recs[int(rng.integers(1 if len(recs) > 1 else 0, len(recs)))]
                if rng.random() < 0.5:
                    c = tuple(np.float32(v) + np.float32(rng.choice([0.0, 1e-7, -1e-7])) for v in c)
            else:
                c, r = (float(rng.uniform(-ext, ext)), float(rng.uniform(0.05, 0.4 * ext)), float(rng.uniform(-ext, ext))), float(rng.uniform(0.03, 0.12) * ext)
            recs.append((tuple(float(v) for v in c), float(r), tuple(float(v) for v in rng.uniform(0.1, 1.0, 3)),
                         int(rng.choice([T.SHADING_LAMBERT] * 4 + [T.SHADING_MIRROR, T.SHADING_GLASS])), 1.5))
        if hostile:      # finite but absurd (anything non-finite or inverted takes the second tree out of use): points, specks, giants, far away
            for _ in range(int(rng.integers(1, 6))):
                j = int(rng.integers(0, len(recs)))
                c, r, kd, sh, ior = recs[j]
                if rng.random() < 0.5: r = float(rng.choice([0.0, 1e-30, 3e-39, 1e-6, 1e4 * ext, 1e18, 1e30]))
                else: c = tuple(float(rng.choice([1e18, -1e18, 1e30, 0.0, 3e-39, 1e6 * ext])) if i == int(rng.integers(0, 3)) else v for i, v in enumerate(c))
                recs[j] = (c, r, kd, sh, ior)
        order = rng.permutation(len(recs))

        def build(b):
            ids = [b.add_sphere(scenes.sphere(*recs[int(j)])) for j in order]
            for i in ids:
                b.build_sphere_instance([i])
            b.rebuild_tlas()
        so = orc.OrcScene(); build(so)
        s = engine.Scene(); build(s); renderer.commit(s)
        cfg = scenes.Config("fz2", 64, 40, 2, (float(rng.uniform(-1, 1)) * ext, float(rng.uniform(0.3, 1.0)) * ext, 2.2 * ext), (0.0, 0.15 * ext, 0.0),
                            max_depth=int(rng.choice([2, 3, 5])))
        po_ = scenes.frame_params(cfg, *H.host_funcs("orc", orc))
        ref, oo = T.alloc_outputs(64, 40)
        ost = orc.render_frame(so.desc(), po_, oo, None)
        pg_ = scenes.frame_params(cfg, *H.host_funcs("hrt"))
        for fl in (0, T.FLAG_STREAMED, T.FLAG_STREAMED | T.FLAG_COUNTERS, T.FLAG_MEGAKERNEL):
            renderer.reset_history()
            got, og = T.alloc_outputs(64, 40)
            st = renderer.render_params(pg_, og, flags=fl)
            bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
            bad = {k: v for k, v in bad.items() if v}
            if not bad and (fl & T.FLAG_COUNTERS) and st.k[1].as_dict() != ost.k[1].as_dict():
                bad = {"counters": 1}
            if bad:
                failures.append((case, fl, bad))
                break
        if len(failures) >= 5:
            break
    assert not failures, "cases that differ from the oracle (case, flags, {array: elements}): %s" % failures


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("hostile", [False, True])
def test_random_many_sphere_scenes_with_sphere_updates(orc, renderer, hostile):
    """256..600 one-sphere instances, then two rounds of moved / resized spheres under a random policy: the second tree keeps its
    topology and follows (refit of its boxes, of its renumbered copies, a new slot map when the tree in use was rebuilt), or stands
    down when a sphere is no longer a regular box (hostile).  Frames in four organisations against the oracle over the arrays the
    device holds after each round."""
    from tests import test_bvh_update_gpu as U
    n_cases = max(4, N_CASES // 20)
    failures = []
    for case in range(n_cases):
        rng = np.random.default_rng(SEED0 + 0x500000 + case + (0x40000 if hostile else 0))
        n = int(rng.integers(256, 600))
        ext = float(rng.choice([2.0, 5.0, 12.0]))
        s = engine.Scene()
        ids = []
        if rng.random() < 0.7:
            ids.append(s.add_sphere(scenes.sphere((0.0, -500.0, 0.0), 500.0, (0.6, 0.6, 0.6))))
        for i in range(n):
            c = (float(rng.uniform(-ext, ext)), float(rng.uniform(0.05, 0.4 * ext)), float(rng.uniform(-ext, ext)))
            ids.append(s.add_sphere(scenes.sphere(c, float(rng.uniform(0.03, 0.12) * ext), tuple(float(v) for v in rng.uniform(0.1, 1.0, 3)),
                                                  int(rng.choice([T.SHADING_LAMBERT] * 4 + [T.SHADING_MIRROR, T.SHADING_GLASS])), 1.5)))
        for i in rng.permutation(len(ids)):
            s.build_sphere_instance([ids[int(i)]])
        s.rebuild_tlas()
        renderer.commit(s)
        arrs = s.arrays()
        cfg = scenes.Config("fz3", 64, 40, 2, (float(rng.uniform(-1, 1)) * ext, float(rng.uniform(0.3, 1.0)) * ext, 2.2 * ext), (0.0, 0.15 * ext, 0.0),
                            max_depth=int(rng.choice([2, 3, 5])))
        po_ = scenes.frame_params(cfg, *H.host_funcs("orc", orc))
        pg_ = scenes.frame_params(cfg, *H.host_funcs("hrt"))
        for rnd in range(2):
            policy = int(rng.choice([T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD, T.REBUILD_AUTO]))
            n_sph = len(arrs["spheres"])
            first = int(rng.integers(0, n_sph)); cnt = int(rng.integers(1, n_sph - first + 1))
            sp = arrs["spheres"].copy()
            for f_ in "XYZ":
                sp["center"][f_][first:first + cnt] += (rng.uniform(-0.3, 0.3, cnt) * ext * 0.2).astype(np.float32)
            sp["radius"][first:first + cnt] *= rng.uniform(0.6, 1.3, cnt).astype(np.float32)
            if rng.random() < 0.3:                    # twins: exactly equal distances on the refitted second tree too
                a_, b_ = int(rng.integers(first, first + cnt)), int(rng.integers(first, first + cnt))
                for f_ in "XYZ": sp["center"][f_][a_] = sp["center"][f_][b_]
                sp["radius"][a_] = sp["radius"][b_]
            if hostile:
                for _ in range(int(rng.integers(1, 3))):
                    i_ = first + int(rng.integers(0, cnt))
                    if rng.random() < 0.5: sp["center"]["XYZ"[int(rng.integers(0, 3))]][i_] = rng.choice(np.array([NAN, INF, 1e30, 1e18, 0.0], np.float32))
                    else: sp["radius"][i_] = rng.choice(np.array([0.0, -0.3, INF, NAN, 1e-30, 1e30], np.float32))
            renderer.update_spheres(first, sp[first:first + cnt], policy)
            arrs["spheres"] = sp
            nodes, idx, inst = U._download(renderer)
            arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = renderer.download_array("blasNodes"), inst, nodes, idx
            desc, keep = T.scene_desc_from_arrays(arrs)
            ref, oo = T.alloc_outputs(64, 40)
            ost = orc.render_frame(desc, po_, oo, None)
            bad = {}
            for fl in (0, T.FLAG_STREAMED, T.FLAG_STREAMED | T.FLAG_COUNTERS, T.FLAG_MEGAKERNEL):
                renderer.reset_history()
                got, og = T.alloc_outputs(64, 40)
                st = renderer.render_params(pg_, og, flags=fl)
                bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
                bad = {k: v for k, v in bad.items() if v}
                if not bad and (fl & T.FLAG_COUNTERS) and st.k[1].as_dict() != ost.k[1].as_dict():
                    bad = {"counters": 1}
                if bad:
                    failures.append((case, rnd, policy, fl, bad))
                    break
            if bad:
                break
        if len(failures) >= 5:
            break
    assert not failures, "cases that differ from the oracle (case, round, policy, flags, {array: elements}): %s" % failures


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("hostile", [False, True])
def test_random_moves_through_the_device_update_path(orc, renderer, hostile):
    """Random scenes, then random instance moves (rigid, scaled, identity, a degenerate one now and then) under a random policy
    (refit / rebuild / auto) on the device: the instance records must be the oracle's, and the frame must be the oracle's frame on
    the tree the device made (downloaded)."""
    from tests import test_bvh_update_gpu as U
    n_cases = max(8, N_CASES // 4)
    failures = []
    for case in range(n_cases):
        seed = SEED0 + 0x200000 + case + (0x40000 if hostile else 0)
        ops, fr = _scene_recipe(seed)
        rng = np.random.default_rng(seed ^ 0xABCDEF)
        if hostile and rng.random() < 0.5:
            ops, fr = _poison(ops, fr, seed)
        so = orc.OrcScene(); _apply(so, ops)
        s = engine.Scene(); _apply(s, ops)
        renderer.commit(s)
        n_inst = len(so.arrays()["instances"])
        if n_inst < 2:
            continue
        ids = sorted(set(int(v) for v in rng.integers(0, n_inst, int(rng.integers(1, min(n_inst, 8) + 1)))))
        xfs = []
        for _ in ids:
            k = rng.random()
            if k < 0.15: xfs.append(T.identity_affine())
            elif k < 0.9: xfs.append(scenes.rotation_affine(str(rng.choice(list("xyz"))), float(rng.uniform(-90, 90)), float(rng.choice([0.5, 0.8, 1.0, 1.0, 1.25])), tuple(float(v) for v in rng.uniform(-0.8, 0.8, 3))))
            else: xfs.append(scenes.rotation_affine("y", 10.0, float(rng.choice([0.0, -1.0, 1e10])), (0.1, 0.2, 0.3)))
            if hostile and rng.random() < 0.3:      # one matrix entry replaced: a shear, a zero column, now and then a NaN / infinity
                m = xfs[-1]
                setattr(m, "m%d%d" % (int(rng.integers(0, 3)), int(rng.integers(0, 4))), float(rng.choice([0.0, 0.7, -2.5, 1e19, 1e-30] + ([NAN, INF] if rng.random() < 0.3 else []))))
        policy = int(rng.choice([T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD, T.REBUILD_AUTO]))
        try:
            renderer.update_instances(ids, xfs, policy)
        except engine.HrtError:
            if policy != T.REBUILD_FORCE_REFIT:
                raise
            renderer.update_instances(ids, xfs, T.REBUILD_FORCE_REBUILD)     # a tree that cannot be refitted says so; rebuild instead
        for i, m in zip(ids, xfs):
            so.set_instance_transform(i, m)
        nodes, idx, inst = U._download(renderer)
        if H.canon(inst).tobytes() != H.canon(so.arrays()["instances"]).tobytes():
            failures.append((seed, "instance records")); continue
        desc = U._desc_with_tlas(so.desc(), nodes, idx, inst)
        cfg = scenes.Config("fz", fr["w"], fr["h"], fr["spp"], fr["origin"], fr["lookat"], max_depth=fr["max_depth"], vfov=fr["vfov"],
                            extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
        po_ = scenes.frame_params(cfg, *H.host_funcs("orc", orc), frame=fr["frame"], rng_lock_noise=fr["lock"])
        ref, oo = T.alloc_outputs(fr["w"], fr["h"])
        ost = orc.render_frame(desc, po_, oo, None)
        pg_ = scenes.frame_params(cfg, *H.host_funcs("hrt"), frame=fr["frame"], rng_lock_noise=fr["lock"])
        for fl in (0, FORCED[case % len(FORCED)]):
            renderer.reset_history()
            got, og = T.alloc_outputs(fr["w"], fr["h"])
            st = renderer.render_params(pg_, og, flags=fl)
            bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
            bad = {k: v for k, v in bad.items() if v}
            if not bad and (fl & T.FLAG_COUNTERS) and st.k[1].as_dict() != ost.k[1].as_dict():
                bad = {"counters": 1}
            if bad:
                failures.append((seed, fl, policy, bad)); break
        if len(failures) >= 5:
            break
    assert not failures, "cases that differ from the oracle: %s" % failures


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("hostile", [False, True])
def test_random_sphere_and_vertex_updates_on_the_device(orc, renderer, hostile):
    """Random scenes, then a random range of spheres moved / resized / recoloured (hrt_scene_update_spheres) or a random range of mesh
    vertices displaced (hrt_scene_update_positions) under a random policy: the device's BLAS boxes and instance bounds must be the
    numpy refit's, and the frame the oracle's frame over the arrays the device now holds."""
    from tests import test_bvh_update_gpu as U
    n_cases = max(8, N_CASES // 4)
    failures = []
    for case in range(n_cases):
        seed = SEED0 + 0x300000 + case + (0x40000 if hostile else 0)
        ops, fr = _scene_recipe(seed)
        rng = np.random.default_rng(seed ^ 0x13579B)
        wild = hostile and rng.random() < 0.33       # as in _poison: non-finite numbers hide the whole scene, so most cases stay finite
        bad_coord = np.array([NAN, INF, -INF, 1e30] if wild else [1e30, -1e30, 1e18, 0.0, 3e-39], np.float32)
        bad_radius = np.array([0.0, -0.3, INF, NAN, 1e-30, 1e30] if wild else [0.0, -0.3, -2.0, 1e-30, 3e-39, 1e30], np.float32)
        s = engine.Scene(); _apply(s, ops)
        renderer.commit(s)
        arrs = s.arrays()
        policy = int(rng.choice([T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD, T.REBUILD_AUTO]))
        n_pos, n_sph = len(arrs["meshPositions"]), len(arrs["spheres"])
        what = "positions" if (n_pos > 0 and rng.random() < 0.5) else "spheres"
        try:
            if what == "spheres":
                first = int(rng.integers(0, n_sph)); cnt = int(rng.integers(1, n_sph - first + 1))
                sp = arrs["spheres"].copy()
                for f_ in "XYZ":
                    sp["center"][f_][first:first + cnt] += rng.uniform(-0.4, 0.4, cnt).astype(np.float32)
                sp["radius"][first:first + cnt] *= rng.uniform(0.6, 1.3, cnt).astype(np.float32)
                sp["albedo"]["Y"][first:first + cnt] = rng.uniform(0.1, 0.9, cnt).astype(np.float32)
                if hostile:
                    for _ in range(int(rng.integers(1, 4))):
                        i_ = first + int(rng.integers(0, cnt))
                        if rng.random() < 0.5: sp["center"]["XYZ"[int(rng.integers(0, 3))]][i_] = rng.choice(bad_coord)
                        else: sp["radius"][i_] = rng.choice(bad_radius)
                renderer.update_spheres(first, sp[first:first + cnt], policy)
                arrs["spheres"] = sp
                with np.errstate(all="ignore"):
                    want_blas, want_inst = U._refit_sphere_blas_numpy(arrs)
            else:
                first = int(rng.integers(0, n_pos)); cnt = int(rng.integers(1, n_pos - first + 1))
                pos = np.stack([arrs["meshPositions"][f_] for f_ in "XYZ"], axis=1)
                pos[first:first + cnt] += rng.uniform(-0.15, 0.15, (cnt, 3)).astype(np.float32)
                if hostile:
                    for _ in range(int(rng.integers(1, 4))):
                        pos[first + int(rng.integers(0, cnt)), int(rng.integers(0, 3))] = rng.choice(bad_coord)
                renderer.update_positions(first, pos[first:first + cnt], policy)
                for k_, f_ in enumerate("XYZ"):
                    arrs["meshPositions"][f_] = pos[:, k_]
                with np.errstate(all="ignore"):
                    want_blas, want_inst = U._refit_blas_numpy(arrs)
        except engine.HrtError as e:
            if "cannot be refitted" in str(e) or "INVALID_STATE" in str(e) or "refit" in str(e):
                continue                                     # the library says when a BLAS layout is not one it maintains
            raise
        got_blas = renderer.download_array("blasNodes")
        nodes, idx, inst = U._download(renderer)
        if policy == T.REBUILD_FORCE_REFIT and (H.canon(got_blas).tobytes() != H.canon(want_blas).tobytes() or H.canon(inst).tobytes() != H.canon(want_inst).tobytes()):
            failures.append((seed, what, policy, "device BLAS / instance bounds differ from the numpy refit")); continue
        # (under the other policies the device may have given a BLAS a new topology: the frame is checked over what it holds now)
        arrs["blasNodes"], arrs["instances"], arrs["tlasNodes"], arrs["tlasInstanceIndices"] = got_blas, inst, nodes, idx
        arrs["triPrimIdx"], arrs["spherePrimIdx"] = renderer.download_array("triPrimIdx"), renderer.download_array("spherePrimIdx")
        desc, keep = T.scene_desc_from_arrays(arrs)
        cfg = scenes.Config("fz", fr["w"], fr["h"], fr["spp"], fr["origin"], fr["lookat"], max_depth=fr["max_depth"], vfov=fr["vfov"],
                            extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
        po_ = scenes.frame_params(cfg, *H.host_funcs("orc", orc), frame=fr["frame"], rng_lock_noise=fr["lock"])
        ref, oo = T.alloc_outputs(fr["w"], fr["h"])
        ost = orc.render_frame(desc, po_, oo, None)
        pg_ = scenes.frame_params(cfg, *H.host_funcs("hrt"), frame=fr["frame"], rng_lock_noise=fr["lock"])
        for fl in (0, FORCED[case % len(FORCED)]):
            renderer.reset_history()
            got, og = T.alloc_outputs(fr["w"], fr["h"])
            st = renderer.render_params(pg_, og, flags=fl)
            bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
            bad = {k: v for k, v in bad.items() if v}
            if not bad and (fl & T.FLAG_COUNTERS) and st.k[1].as_dict() != ost.k[1].as_dict():
                bad = {"counters": 1}
            if bad:
                failures.append((seed, what, fl, policy, bad)); break
        if len(failures) >= 5:
            break
    assert not failures, "cases that differ: %s" % failures


@pytest.mark.timeout(1800)
def test_random_scenes_over_several_device_slots(orc, hrt_lib):
    """The same random scenes through ONE context over 2, 3 or 5 device slots (row strips dealt round-robin, per-slot gather, tile
    exchange under reuse), frame sizes with ragged and missing strips: the assembled frame is the oracle's."""
    n_cases = max(6, N_CASES // 6)
    multi = {n: engine.RTRenderer([0] * n) for n in (2, 3, 5)}
    failures = []
    try:
        for case in range(n_cases):
            seed = SEED0 + 0x400000 + case
            ops, fr = _scene_recipe(seed)
            rng = np.random.default_rng(seed ^ 0x2468AC)
            n = int(rng.choice([2, 3, 5]))
            r = multi[n]
            w, h = int(rng.choice([24, 40, 57])), int(rng.choice([5, 8, 13, 24, 33, 47]))
            so = orc.OrcScene(); _apply(so, ops)
            s = engine.Scene(); _apply(s, ops)
            r.commit(s); r.reset_history()
            cfg = scenes.Config("fz", w, h, fr["spp"], fr["origin"], fr["lookat"], max_depth=fr["max_depth"], vfov=fr["vfov"],
                                extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
            fl = int(rng.choice([0, T.FLAG_STREAMED, T.FLAG_MEGAKERNEL, T.FLAG_COUNTERS]))
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
                ost = orc.render_frame(so.desc(), po_, oo, po)
                got, og = T.alloc_outputs(w, h)
                st = r.render_params(pg_, og, flags=fl)
                bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
                bad = {k: v for k, v in bad.items() if v}
                if not bad and (fl & T.FLAG_COUNTERS) and any(st.k[i].as_dict() != ost.k[i].as_dict() for i in range(2)):
                    bad = {"counters": 1}
                if bad:
                    failures.append((seed, n, w, h, fl, f, bad)); break
            if len(failures) >= 5:
                break
    finally:
        for r in multi.values():
            r.close()
    assert not failures, "cases that differ from the oracle (seed, slots, w, h, flags, frame, {array: elements}): %s" % failures


@pytest.mark.timeout(1800)
def test_random_call_sequences(orc, hrt_lib):
    """Random SEQUENCES of calls on one renderer -- commit another scene, resize, synchronous frames with outputs in any kernel
    organisation, runs of asynchronous frames, instance moves on the device, history resets, workspace limits -- with ReSTIR reuse on
    for most frames, so that every frame depends on the reservoirs every earlier call left behind.  The oracle keeps the same two
    reservoir sets (ping-pong by frame parity, kept across commits and across resizes to the same pixel count, zeroed by a reset or a new length)."""
    from tests import test_bvh_update_gpu as U
    n_seq = max(4, N_CASES // 10)
    failures = []
    for seq in range(n_seq):
        rng = np.random.default_rng(SEED0 + 0x500000 + seq)
        r = engine.RTRenderer([0])
        try:
            w = h = 0
            so = desc = None
            moved_spheres, keep2 = False, None
            A = B = None
            frame = int(rng.integers(0, 50))
            log = []
            for step in range(int(rng.integers(8, 20))):
                op = rng.choice(["commit", "frame", "frame", "frame", "async", "move", "spheres", "reset", "limit", "resize"]) if so is not None else "commit"
                if op == "move" and moved_spheres: op = "frame"       # (the oracle's scene object no longer holds the spheres the device has)
                log.append(str(op))
                if op == "commit":
                    ops, fr = _scene_recipe(int(rng.integers(0, 2 ** 31)))
                    moved_spheres, keep2 = False, None
                    if rng.random() < 0.3:      # 260..400 one-sphere instances: the library keeps a second tree for such a scene, across the calls below
                        ext = float(rng.choice([2.0, 5.0]))
                        ops = [("sphere", (0.0, -500.0, 0.0), 500.0, (0.6, 0.6, 0.6), T.SHADING_LAMBERT, 1.0, True)]
                        for _ in range(int(rng.integers(260, 400))):
                            ops.append(("sphere", (float(rng.uniform(-ext, ext)), float(rng.uniform(0.05, 0.4 * ext)), float(rng.uniform(-ext, ext))), float(rng.uniform(0.03, 0.12) * ext),
                                        tuple(float(v) for v in rng.uniform(0.1, 1.0, 3)), int(rng.choice([T.SHADING_LAMBERT] * 4 + [T.SHADING_MIRROR, T.SHADING_GLASS])), 1.5, True))
                        fr = dict(fr, origin=(float(rng.uniform(-1, 1)) * ext, float(rng.uniform(0.3, 1.0)) * ext, 2.2 * ext), lookat=(0.0, 0.15 * ext, 0.0), vfov=60.0)
                        log[-1] += " many"
                    so = orc.OrcScene(); _apply(so, ops)
                    s = engine.Scene(); _apply(s, ops)
                    r.commit(s); desc = so.desc()
                    cfg_fr = fr
                    if w == 0: w, h = fr["w"], fr["h"]
                elif op == "resize":
                    w, h = int(rng.choice([24, 32, 48])), int(rng.choice([16, 24, 40]))       # takes effect at the next frame
                    log[-1] += " %dx%d" % (w, h)
                elif op == "reset":
                    r.reset_history()
                    if A is not None: A, B = H.new_reservoirs(len(A["res_m"]), 1), H.new_reservoirs(len(A["res_m"]), 1)
                elif op == "limit":
                    r.set_workspace_limit(int(rng.choice([0, w * h + 3, 2 * w * h + 1, 10 ** 9])))
                elif op == "spheres":
                    arrs2 = so.arrays() if keep2 is None else keep2[0]
                    sp = arrs2["spheres"].copy()
                    n_sph = len(sp)
                    first = int(rng.integers(0, n_sph)); cnt = int(rng.integers(1, n_sph - first + 1))
                    for f_ in "XYZ":
                        sp["center"][f_][first:first + cnt] += rng.uniform(-0.2, 0.2, cnt).astype(np.float32)
                    sp["radius"][first:first + cnt] *= rng.uniform(0.7, 1.2, cnt).astype(np.float32)
                    try:
                        r.update_spheres(first, sp[first:first + cnt], int(rng.choice([T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD, T.REBUILD_AUTO])))
                    except engine.HrtError as e:
                        if "refit" not in str(e): raise
                        log[-1] += " (refused)"
                        continue
                    arrs2 = dict(arrs2); arrs2["spheres"] = sp
                    nodes, idx, inst = U._download(r)
                    arrs2["blasNodes"], arrs2["instances"], arrs2["tlasNodes"], arrs2["tlasInstanceIndices"] = r.download_array("blasNodes"), inst, nodes, idx
                    arrs2["triPrimIdx"], arrs2["spherePrimIdx"] = r.download_array("triPrimIdx"), r.download_array("spherePrimIdx")
                    desc, k2 = T.scene_desc_from_arrays(arrs2)
                    keep2 = (arrs2, k2)
                    moved_spheres = True
                elif op == "move":
                    n_inst = len(so.arrays()["instances"])
                    ids = sorted(set(int(v) for v in rng.integers(0, n_inst, int(rng.integers(1, min(n_inst, 4) + 1)))))
                    xfs = [scenes.rotation_affine(str(rng.choice(list("xyz"))), float(rng.uniform(-60, 60)), float(rng.choice([0.7, 1.0, 1.0])), tuple(float(v) for v in rng.uniform(-0.5, 0.5, 3))) for _ in ids]
                    r.update_instances(ids, xfs, int(rng.choice([T.REBUILD_FORCE_REBUILD, T.REBUILD_AUTO])))
                    for i, m in zip(ids, xfs):
                        so.set_instance_transform(i, m)
                    nodes, idx, inst = U._download(r)
                    keep = (nodes, idx, inst)
                    desc = U._desc_with_tlas(so.desc(), nodes, idx, inst)
                else:
                    fr = cfg_fr
                    n_frames = 1 if op == "frame" else int(rng.integers(2, 5))
                    reuse = bool(rng.random() < 0.75)
                    fl = int(rng.choice([0, 0, T.FLAG_MEGAKERNEL, T.FLAG_STREAMED, T.FLAG_STREAMED | T.FLAG_REFERENCE_LAYOUT, T.FLAG_COUNTERS]))
                    spp = int(rng.integers(1, 4))
                    cfg = scenes.Config("sq", w, h, spp, fr["origin"], fr["lookat"], max_depth=fr["max_depth"], vfov=fr["vfov"],
                                        extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
                    if A is None or len(A["res_m"]) != w * h:
                        # EnsureLength reallocates (and this library zeroes) on another LENGTH only (Framebuffer.cs:60-66): a frame of 48x16
                        # after one of 32x24 keeps every array, stale history included
                        A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
                    for j in range(n_frames):
                        frame += int(rng.integers(1, 3))
                        last = j == n_frames - 1
                        po_ = scenes.frame_params(cfg, *H.host_funcs("orc", orc), frame=frame, reuse=reuse, rng_lock_noise=fr["lock"])
                        pg_ = scenes.frame_params(cfg, *H.host_funcs("hrt"), frame=frame, reuse=reuse, rng_lock_noise=fr["lock"])
                        prev, cur = (B, A) if (frame & 1) == 0 else (A, B)
                        ref, oo = T.alloc_outputs(w, h)
                        for k, a in cur.items():
                            ref[k] = a; setattr(oo, k, a.ctypes.data)
                        po = T.Outputs()
                        for k, a in prev.items():
                            setattr(po, k, a.ctypes.data)
                        orc.render_frame(desc, po_, oo, po)
                        if op == "async" and not last:
                            r.render_params(pg_, None, flags=(fl & ~T.FLAG_COUNTERS) | T.FLAG_NO_SYNC)
                            continue
                        got, og = T.alloc_outputs(w, h)
                        r.render_params(pg_, og, flags=fl)
                        bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
                        bad = {k: v for k, v in bad.items() if v}
                        if bad:
                            failures.append((seq, step, log[-6:], fl, reuse, bad)); break
                    log[-1] += " x%d fl%d reuse%d" % (n_frames, fl, reuse)
                if failures and failures[-1][0] == seq:
                    break
        finally:
            r.close()
        if len(failures) >= 4:
            break
    assert not failures, "sequences that diverge from the oracle (sequence, step, last calls, flags, reuse, {array: elements}): %s" % failures
