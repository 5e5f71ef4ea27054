"""The treelet cut of big mesh BLASes (csrc/hrt_treelets.hpp), host side, through the test hook hrt_debug_treelets.

What the LDS-staged walker (HRT_FLAG_TREELETS) needs from it, checked structurally on the arrays:
  * treelets are disjoint subtrees of the packed (walk-order) BLAS whose nodes AND triangle records are contiguous ranges
    and fit the byte limit; small ones are not cut out;
  * a skip-link walk over reduced tree + treelets visits the nodes of the packed BLAS in the same order with the same boxes
    as the plain walk (the portal's box IS the root's box, the walk goes on at root + 1 and comes back at the root's skip);
  * the reduced array is ordered top first (what a workgroup keeps in LDS is a prefix)."""
import ctypes as C

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes

K_END = 0x0FFFFFFF
PORTAL = 0x40000000


def _treelets(lib, builder, limits):
    s = engine.Scene()
    builder(s)
    d = s.desc()
    cnt = (C.c_int64 * 3)()
    assert lib.hrt_debug_treelets(C.byref(d), *limits, None, None, None, None, None, cnt) == 0
    nB, nR, nT = cnt[0], cnt[1], cnt[2]
    blas = np.zeros((nB, 8), np.float32); red = np.zeros((max(nR, 1), 8), np.float32)
    orig = np.zeros(max(nR, 1), np.int32); tl = np.zeros((max(nT, 1), 8), np.int32); ror = np.full(nB, -1, np.int32)
    assert lib.hrt_debug_treelets(C.byref(d), *limits, blas.ctypes.data, red.ctypes.data, orig.ctypes.data, tl.ctypes.data, ror.ctypes.data, cnt) == 0
    return s, blas, red[:nR], orig[:nR], tl[:nT], ror


def _bits(a):
    return np.ascontiguousarray(a).view(np.int32)


def _hit(rec, o, inv):
    with np.errstate(all="ignore"):
        t1 = (rec[0:3] - o) * inv; t2 = (rec[4:7] - o) * inv
        tmn = np.minimum(t1, t2).max(); tmx = np.maximum(t1, t2).min()
    return bool(tmx >= max(tmn, np.float32(0.001)))


def _walk_plain(blas, lo, hi, o, inv):
    seq, cur = [], lo
    link, skw = _bits(blas[:, 3]), _bits(blas[:, 7])
    while cur < hi:
        seq.append(cur)
        sk = int(skw[cur]) & K_END; cnt = (int(skw[cur]) >> 28) & 15
        if not _hit(blas[cur], o, inv) or cnt > 0: cur = sk
        else: cur = int(link[cur]) & K_END
    return seq


def _walk_reduced(blas, red, orig, tl, root, o, inv):
    seq, cur = [], root
    link, skw = _bits(blas[:, 3]), _bits(blas[:, 7])
    rlink, rskw = _bits(red[:, 3]), _bits(red[:, 7])
    while cur != K_END:
        seq.append(int(orig[cur]))
        sk = int(rskw[cur]) & K_END; cnt = (int(rskw[cur]) >> 28) & 15
        assert np.array_equal(red[cur][[0, 1, 2, 4, 5, 6]], blas[orig[cur]][[0, 1, 2, 4, 5, 6]])       # the record carries the node's own box
        if not _hit(red[cur], o, inv) or cnt > 0: cur = sk; continue
        l = int(rlink[cur])
        if l & PORTAL:
            t = tl[l & K_END]
            assert t[0] == orig[cur] and t[4] == sk
            c = t[0] + 1
            while c < t[1]:
                seq.append(int(c))
                s2 = int(skw[c]) & K_END; n2 = (int(skw[c]) >> 28) & 15
                if not _hit(blas[c], o, inv) or n2 > 0: c = s2
                else: c = int(link[c]) & K_END
            cur = sk
        else:
            cur = l & K_END
    return seq


CASES = {
    "blob_64x64": (lambda b: scenes.build_config4(b, 64, 64), (4096, 7, 64)),
    "terrain_96": (lambda b: scenes.build_config5(b, 96), (6000, 15, 64)),
    "textured_test_scene": (scenes.build_textured_test_scene, (1024, 3, 8)),
    "rotated_scaled_instances": (scenes.build_rotated_instances_scene, (1024, 3, 8)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_reduced_tree_plus_treelets_is_the_same_walk(hooks_lib, name):
    builder, limits = CASES[name]
    s, blas, red, orig, tl, ror = _treelets(hooks_lib, builder, limits)
    inst = s.arrays()["instances"]
    meshes = [(int(i["blasRoot"]), int(i["blasRoot"]) + int(i["blasNodeCount"])) for i in inst if i["type"] == 2 and ror[int(i["blasRoot"])] >= 0]
    if name in ("blob_64x64", "terrain_96"):
        assert meshes and len(tl) >= 4
    if not meshes:
        assert len(tl) == 0 and len(red) == 0
        return
    link, skw = _bits(blas[:, 3]), _bits(blas[:, 7])
    # treelets: disjoint node ranges inside one mesh, contiguous triangle ranges that are exactly their leaves', within the limits
    used = np.zeros(len(blas), bool)
    for t in tl:
        lo, hi, tlo, thi = int(t[0]), int(t[1]), int(t[2]), int(t[3])
        assert any(a <= lo and hi <= b for a, b in meshes) and hi - lo >= limits[1]
        assert not used[lo:hi].any(); used[lo:hi] = True
        assert (int(skw[lo]) & K_END) in (hi, K_END) or (int(skw[lo]) & K_END) >= hi          # the subtree of lo is [lo, hi)
        cnts = (skw[lo:hi] >> 28) & 15
        leaves = np.flatnonzero(cnts > 0) + lo
        firsts = link[leaves]
        assert firsts.min() == tlo and (firsts + cnts[leaves - lo]).max() == thi and cnts.sum() == thi - tlo
        assert 32 * (hi - lo) + 48 * (thi - tlo) <= limits[0]
    # every node of a mesh is either inside a treelet or copied by exactly one reduced record (a treelet's root: by its portal)
    covered = used.copy()
    for t in tl: covered[t[0]] = False
    for a, b in meshes:
        outside = np.flatnonzero(~covered[a:b]) + a
        assert sorted(orig[np.isin(orig, outside)].tolist()) == outside.tolist()
    # top first: subtree sizes never grow along the reduced array
    sub = np.array([(int(skw[o]) & K_END if (int(skw[o]) & K_END) != K_END else next(b for a, b in meshes if a <= o < b)) - o for o in orig])
    tri_of = lambda o, e: int(((skw[o:e] >> 28) & 15).sum())
    size = np.array([32 * n + 48 * tri_of(int(o), int(o) + int(n)) for o, n in zip(orig, sub)])
    assert np.all(size[:-1] >= size[1:])
    # the walks
    rng = np.random.default_rng(3)
    for a, b in meshes:
        lo3, hi3 = blas[a][0:3], blas[a][4:7]
        for _ in range(60):
            o = (lo3 + (hi3 - lo3) * rng.uniform(-0.3, 1.3, 3)).astype(np.float32)
            d = rng.standard_normal(3).astype(np.float32); d /= np.linalg.norm(d)
            inv = (np.float32(1.0) / np.where(d != 0, d, np.float32(1e-8))).astype(np.float32)
            assert _walk_plain(blas, a, b, o, inv) == _walk_reduced(blas, red, orig, tl, int(ror[a]), o, inv)


def test_small_or_unfit_meshes_get_no_treelets(hooks_lib):
    # shipped limits: a mesh below 4096 nodes is walked as before
    s, blas, red, orig, tl, ror = _treelets(hooks_lib, lambda b: scenes.build_config4(b, 16, 16), (0, 0, 0))
    assert len(tl) == 0 and len(red) == 0
    # a scene without meshes
    s, blas, red, orig, tl, ror = _treelets(hooks_lib, scenes.build_config2, (1024, 3, 8))
    assert len(tl) == 0
