"""Host half of the SECOND tree of scenes of many one-sphere instances (csrc/hrt_runtime.hip: host_sah_topology,
reorder_second_tree), through the two host-only test hooks of include/hip_raytrace.h -- no GPU.  The pictures cannot depend on
either (the GPU tests compare them with the oracle); what is checked here is that the arrays are the trees the
walkers assume: a binary tree in walk order over every instance exactly once, and renumberings that are permutations of the same
records with consistent links and the nearer child first."""
import ctypes as C

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes

K_END = 0x0FFFFFFF
ERR_INVALID_ARG = -1          # HRT_ERR_INVALID_ARG (include/hip_raytrace.h)


def _instances(builder):
    s = engine.Scene()
    builder(s)
    return s.arrays()["instances"]


def _topology(lib, inst):
    n = len(inst)
    order = np.zeros(n, np.int32)
    link, skip, count, parent = (np.zeros(2 * n, np.int32) for _ in range(4))
    nn = C.c_int32(0)
    lib.hrt_debug_second_tree_topology.argtypes = [C.c_void_p] + [C.c_int32] + [C.c_void_p] * 6
    rc = lib.hrt_debug_second_tree_topology(inst.ctypes.data, n, order.ctypes.data, link.ctypes.data, skip.ctypes.data, count.ctypes.data,
                                            parent.ctypes.data, C.byref(nn))
    assert rc == 0
    k = nn.value
    return order, link[:k], skip[:k], count[:k], parent[:k]


def _many(n, seed, ext=5.0, twins=False):
    def build(b):
        rng = np.random.default_rng(seed)
        ids = [b.add_sphere(scenes.sphere((0.0, -500.0, 0.0), 500.0, (0.6, 0.6, 0.6)))]
        for i in range(n):
            c = (float(rng.uniform(-ext, ext)), float(rng.uniform(0.05, 0.4 * ext)), float(rng.uniform(-ext, ext)))
            if twins and i % 3 == 0:
                c = (1.0, 0.5, -1.0)                         # a third of the spheres in one place: the bins cannot split them
            ids.append(b.add_sphere(scenes.sphere(c, float(rng.uniform(0.03, 0.12) * ext), (0.5, 0.5, 0.5))))
        for i in ids:
            b.build_sphere_instance([i])
        b.rebuild_tlas()
    return build


CASES = {"three": _many(2, 1), "300": _many(299, 2), "2000": _many(1999, 3, ext=12.0), "twins": _many(600, 4, twins=True)}


def _check_topology(n, order, link, skip, count, parent):
    nT = len(link)
    assert sorted(order.tolist()) == list(range(n)), "every instance in exactly one slot"
    leaves = np.flatnonzero(count > 0)
    assert nT == 2 * len(leaves) - 1 and count.max() <= 4 and count.sum() == n
    assert parent[0] == -1 and skip[0] == K_END
    slot = 0
    for i in range(nT):
        end = nT if skip[i] == K_END else int(skip[i])
        assert i < end <= nT
        if count[i] > 0:
            assert end == i + 1 and link[i] == slot, "leaves own consecutive slots in walk order"
            slot += int(count[i])
        else:
            l = int(link[i]); r = nT if skip[l] == K_END else int(skip[l])
            assert l == i + 1 and i < r < end and parent[l] == i and parent[r] == i, "first child follows, second starts where the first ends"
            assert (K_END if end == nT else end) == skip[r] or (skip[r] == K_END and end == nT)


@pytest.mark.parametrize("name", list(CASES))
def test_topology_is_a_walk_order_binary_tree(hooks_lib, name):
    inst = _instances(CASES[name])
    order, link, skip, count, parent = _topology(hooks_lib, inst)
    _check_topology(len(inst), order, link, skip, count, parent)


def _inlined(inst, order, link, skip, count):
    """What the device lays out from a topology (hrt_bvh.hip: k_refit + k_derive): node boxes = unions, every leaf followed by one
    record per instance (count field 15, link = slot, skip = next record)."""
    nT = len(link)
    lo = np.stack([inst["worldBoundsMin"][f] for f in "XYZ"], 1); hi = np.stack([inst["worldBoundsMax"][f] for f in "XYZ"], 1)
    nlo, nhi = np.zeros((nT, 3), np.float32), np.zeros((nT, 3), np.float32)
    for i in range(nT - 1, -1, -1):
        if count[i] > 0:
            ids = order[link[i]:link[i] + count[i]]
            nlo[i], nhi[i] = lo[ids].min(0), hi[ids].max(0)
        else:
            l = int(link[i]); r = int(skip[l])
            nlo[i], nhi[i] = np.minimum(nlo[l], nlo[r]), np.maximum(nhi[l], nhi[r])
    before = np.concatenate([[0], np.cumsum(count)])[:nT]              # slots before node i = records inserted before it
    at = np.arange(nT) + before
    nX = nT + int(count.sum())
    X = np.zeros((nX, 8), np.float32)
    W = X.view(np.int32)
    for i in range(nT):
        a = int(at[i]); skx = K_END if skip[i] == K_END else int(at[skip[i]])
        X[a, 0:3], X[a, 4:7] = nlo[i], nhi[i]
        W[a, 7] = skx | (int(count[i]) << 28)
        W[a, 3] = int(link[i]) if count[i] > 0 else int(at[link[i]])
        for j in range(int(count[i])):
            ii = order[link[i] + j]
            X[a + 1 + j, 0:3], X[a + 1 + j, 4:7] = lo[ii], hi[ii]
            W[a + 1 + j, 3] = int(link[i]) + j
            nxt = a + 2 + j if j + 1 < count[i] else skx
            W[a + 1 + j, 7] = int(np.array(nxt | (15 << 28), np.uint32).view(np.int32))
    return X


def _reorder(lib, X, sign, base, inlined):
    out = np.zeros_like(X); frm = np.zeros(len(X), np.int32)
    sg = np.array(sign, np.int32)
    lib.hrt_debug_second_tree_reorder.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    rc = lib.hrt_debug_second_tree_reorder(X.ctypes.data, len(X), sg.ctypes.data, base, 1 if inlined else 0, out.ctypes.data, frm.ctypes.data)
    return rc, out, frm


@pytest.mark.parametrize("name", ["300", "twins"])
def test_renumberings_are_the_same_tree_near_child_first(hooks_lib, name):
    inst = _instances(CASES[name])
    order, link, skip, count, parent = _topology(hooks_lib, inst)
    X = _inlined(inst, order, link, skip, count)
    nX = len(X)
    for sign in [(1, 0, 1), (-1, 0, 1), (1, 0, -1), (-1, 0, -1), (1, 1, 1), (-1, -1, -1), (0, 0, 0)]:
        base = 1000
        rc, out, frm = _reorder(hooks_lib, X, sign, base, True)
        assert rc == 0
        assert sorted(frm.tolist()) == list(range(nX)), "a permutation of the records"
        W, Wo = X.view(np.int32), out.view(np.int32)
        assert np.array_equal(out[:, [0, 1, 2, 4, 5, 6]], X[frm][:, [0, 1, 2, 4, 5, 6]]), "boxes travel with their records"
        assert np.array_equal((Wo[:, 7].view(np.uint32) >> 28), (W[frm, 7].view(np.uint32) >> 28)), "so do the counts"
        # the walk that enters everything visits every record once, in index order; the walk that misses the root ends at once
        cur, seen = base, 0
        while cur != K_END:
            i = cur - base
            assert i == seen, "walk order"
            seen += 1
            c = int(np.uint32(Wo[i, 7]) >> 28)
            sk = int(Wo[i, 7]) & K_END
            if c == 15: cur = sk                                   # instance record: on to the next record
            elif c > 0: cur = base + i + 1                          # leaf: its instance records follow
            else:
                cur = int(Wo[i, 3]) & K_END
                assert cur == base + i + 1
                # near child first along the axis that separates the two children most, for rays of these signs
                a, b = i + 1, (int(Wo[i + 1, 7]) & K_END) - base
                ca, cb = 0.5 * (out[a, 0:3] + out[a, 4:7]), 0.5 * (out[b, 0:3] + out[b, 4:7])
                ax = int(np.argmax(np.abs(ca - cb)))
                if sign[ax] > 0: assert ca[ax] <= cb[ax]
                elif sign[ax] < 0: assert ca[ax] >= cb[ax]
                else: assert frm[a] < frm[b], "the builder's order stays where the signs say nothing"
        assert seen == nX
        assert (int(Wo[0, 7]) & K_END) == K_END
    # sign (0, 0, 0) is the identity
    rc, out, frm = _reorder(hooks_lib, X, (0, 0, 0), 0, True)
    assert rc == 0 and np.array_equal(frm, np.arange(nX)) and out.tobytes() == X.tobytes()


def test_reorder_rejects_what_is_not_such_a_tree(hooks_lib):
    inst = _instances(CASES["300"])
    order, link, skip, count, parent = _topology(hooks_lib, inst)
    X = _inlined(inst, order, link, skip, count)
    bad = X.copy(); bad.view(np.int32)[0, 3] = 5                     # the root's first child is not the next record
    assert _reorder(hooks_lib, bad, (1, 0, 1), 0, True)[0] == ERR_INVALID_ARG
    bad = X.copy(); bad.view(np.int32)[1, 7] = 3                     # a skip link that points backwards
    assert _reorder(hooks_lib, bad, (1, 0, 1), 0, True)[0] == ERR_INVALID_ARG
    assert _reorder(hooks_lib, X, (1, 0, 1), 0, False)[0] == ERR_INVALID_ARG      # inlined records where the plain layout has none
