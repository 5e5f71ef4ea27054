"""Pins the CPU oracle: integer-exact RNG known answers, closed-form
geometry, BVH-vs-brute-force, accuracy of the shared math, the .NET sort restatement, and the
committed golden fixtures.  The reference holds no tests or fixtures for this path,
so these independent checks are what the oracle stands on ("parity unpinned" vs the C# binary)."""
import json
import math
import os

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, scenes
from tests import helpers as H

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

# CreateFromPixel(px,py,frame,sample,salt=0xC0FFEE,lockNoise)
RNG_KAT = [
    ((0, 0, 0, 0, 0), 0xE582D06F, (0x4E629AA8, 0xBBC51253, 0x2860AC14), (0.385172367, 0.769810855, 0.377625704)),
    ((5, 7, 3, 1, 0), 0x0D7C6447, (0xBF6C531D, 0x94F2DADA, 0x3A965CCE), (0.423143208, 0.948651910, 0.587353587)),
    ((5, 7, 3, 1, 12345), 0xA884BE81, (0xD5C4A44B, 0x68F1794D, 0xBCC7AE81), (0.768131912, 0.943257153, 0.780006468)),
    ((1919, 1079, 0, 3, 0), 0x04F4E58F, (0x90EC7C0D, 0xF2D7A4DB, 0xCFD4385D), (0.923767865, 0.842359245, 0.828985035)),
]


@pytest.mark.parametrize("args,seed,us,fs", RNG_KAT)
def test_rng_known_answers(orc, args, seed, us, fs):
    u, f = orc.rng_kat(*args)
    assert u[0] == seed
    assert tuple(u[1:]) == us
    assert np.allclose(f, fs, rtol=0, atol=1e-9)


def _py_seed(px, py, frame, sample, salt, lock):
    """Independent transcription of RNG.CreateFromPixel (RTUtils.cs:116-137) with Python ints."""
    M32, M64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF
    rotl = lambda v, r: ((v << (r & 31)) | (v >> ((32 - r) & 31))) & M32

    def hash32(x):
        x ^= x >> 17; x = x * 0xED5AD4BB & M32
        x ^= x >> 11; x = x * 0xAC4C1B51 & M32
        x ^= x >> 15; x = x * 0x31848BAB & M32
        return x ^ (x >> 14)

    def splitmix32(x):
        x = (x + 0x9E3779B97F4A7C15) & M64
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
        x ^= x >> 31
        return (x ^ (x >> 32)) & M32

    def pcg(x):
        x ^= x >> 16; x = x * 0x7FEB352D & M32
        x ^= x >> 15; x = x * 0x846CA68B & M32
        return x ^ (x >> 16)

    px &= M32; py &= M32
    f = 0 if lock != 0 else frame & M32
    ln = lock & M32
    m0 = (hash32(ln) ^ (ln * 0x1B873593 & M32)) if lock != 0 else 0
    m1 = (rotl(ln, 7) * 0x85EBCA6B & M32) if lock != 0 else 0
    l0a = px ^ 0xB5297A4D
    l0b = ((py * 0x68E31DA4) & M32) ^ ((f * 0x9E3779B1 + 0x85EBCA6B) & M32) ^ m0
    l1a = ((sample ^ 0xC2B2AE35) + rotl(px, 16)) & M32
    l1b = (((salt ^ 0x27D4EB2F) + rotl(py, 8)) & M32) ^ m1
    s0 = splitmix32(((l0a << 32) | l0b) ^ 0xD1B54A32D192ED03)
    s1 = splitmix32(((l1a << 32) | l1b) ^ 0x94D049BB133111EB)
    return pcg(s0 ^ ((rotl(s1, 13) + 0x9E3779B1) & M32)) | 1


def test_rng_seed_matches_python_transcription(orc):
    rng = np.random.default_rng(7)
    for _ in range(300):
        px, py = int(rng.integers(0, 4096)), int(rng.integers(0, 4096))
        frame, sample = int(rng.integers(0, 1000)), int(rng.integers(0, 256))
        lock = int(rng.integers(-2 ** 31, 2 ** 31)) if rng.random() < 0.5 else 0
        u, _ = orc.rng_kat(px, py, frame, sample, lock)
        assert u[0] == _py_seed(px, py, frame, sample, 0xC0FFEE, lock)


def test_xorshift_stream(orc):
    out = np.zeros(64, np.uint32)
    orc.lib().orc_rng_stream(0x9E3779B9, 64, out.ctypes.data)
    g = scenes.XorShift32(0x9E3779B9)
    assert [int(v) for v in out] == [g.next_u() for _ in range(64)]
    orc.lib().orc_rng_stream(0, 1, out.ctypes.data)          # seed 0 -> 1 (RTUtils.cs:28)
    g = scenes.XorShift32(1)
    assert int(out[0]) == g.next_u()


def test_pack_rgba8_and_hash(orc):
    assert orc.lib().orc_pack_rgba8(1.0, 0.5, 0.0) & 0xFFFFFFFF == 0xFFFF7F00   # SURVEY 8c KAT
    assert orc.lib().orc_pack_rgba8(2.0, -1.0, 0.25) & 0xFFFFFFFF == 0xFFFF003F            # clamp01, (int)(255.99*c)

    def h(x):
        M = 0xFFFFFFFF
        x ^= x >> 17; x = x * 0xED5AD4BB & M
        x ^= x >> 11; x = x * 0xAC4C1B51 & M
        x ^= x >> 15; x = x * 0x31848BAB & M
        return x ^ (x >> 14)
    for a, b, c in [(0, 0, 0xB31F5AB1), (12345, 7, 0xB31F5AB1), (2073599, 3, 0xB31F5AB1)]:
        assert orc.lib().orc_hash3(a, b, c) == h(a ^ h(b ^ h(c)))


def test_no_fma_contraction(orc):
    assert orc.lib().orc_fma_contracted() == 0


def _ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 1e-300)


def test_shared_math_accuracy(orc):
    """hrt_math.h transcendental definitions are accurate to a few ulp on the ranges the path uses."""
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 2 * math.pi, 200000).astype(np.float32)
    assert np.max(np.abs(orc.math_eval("sin", x).astype(np.float64) - np.sin(x.astype(np.float64)))) < 2.5e-7
    assert np.max(np.abs(orc.math_eval("cos", x).astype(np.float64) - np.cos(x.astype(np.float64)))) < 2.5e-7
    x = rng.uniform(0.01, 1.45, 100000).astype(np.float32)
    assert np.max(_ulp_err(orc.math_eval("tan", x), np.tan(x.astype(np.float64)))) < 4
    x = rng.uniform(-100, 100, 100000).astype(np.float32)
    assert np.max(_ulp_err(orc.math_eval("atan", x), np.arctan(x.astype(np.float64)))) < 4
    x = rng.uniform(-1, 1, 100000).astype(np.float32)
    assert np.max(np.abs(orc.math_eval("acos", x).astype(np.float64) - np.arccos(x.astype(np.float64)))) < 6e-7
    y = rng.standard_normal(100000).astype(np.float32)
    x = rng.standard_normal(100000).astype(np.float32)
    assert np.max(np.abs(orc.math_eval("atan2", y, x).astype(np.float64) - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 1e-6
    x = (np.abs(rng.standard_normal(100000)) + 1e-3).astype(np.float32)
    want = (np.float32(1.0) / np.sqrt(x)).astype(np.float32)        # 1/Sqrt with two roundings
    assert np.array_equal(orc.math_eval("rsqrt", x), want)


def test_min_max_semantics(orc):
    """minNum/maxNum with -0 < +0 (v_min_f32 / v_max_f32 / PTX min.f32)."""
    nan = np.float32("nan")
    a = np.array([1, 2, nan, 5, nan, 0.0, -0.0, -0.0], np.float32)
    b = np.array([2, 1, 3, nan, nan, -0.0, 0.0, -0.0], np.float32)
    mn = orc.math_eval("fmin", a, b); mx = orc.math_eval("fmax", a, b)
    assert list(mn[:4]) == [1, 1, 3, 5] and np.isnan(mn[4])
    assert list(mx[:4]) == [2, 2, 3, 5] and np.isnan(mx[4])
    assert np.signbit(mn[5]) and np.signbit(mn[6]) and np.signbit(mn[7])
    assert not np.signbit(mx[5]) and not np.signbit(mx[6]) and np.signbit(mx[7])
    f2i = orc.math_eval("f2i", np.array([1.9, -1.9, np.nan, 3e9, -3e9, 2147483520.0], np.float32)).view(np.int32)
    assert list(f2i) == [1, -1, -2 ** 31, -2 ** 31, -2 ** 31, 2147483520]
    r = orc.math_eval("round", np.array([0.5, 1.5, 2.5, -0.5, -1.5], np.float32))
    assert list(r) == [0, 2, 2, -0.0, -2]


def _py_introsort(keys_idx, key):
    """Independent Python transcription of .NET 8 ArraySortHelper<T>.IntrospectiveSort."""
    k = list(keys_idx)
    cmp = lambda a, b: -1 if key[a] < key[b] else (1 if key[a] > key[b] else 0)

    def sig(lo, i, j):
        if cmp(k[lo + i], k[lo + j]) > 0:
            k[lo + i], k[lo + j] = k[lo + j], k[lo + i]

    def down(lo, i, n):
        d = k[lo + i - 1]
        while i <= n >> 1:
            c = 2 * i
            if c < n and cmp(k[lo + c - 1], k[lo + c]) < 0:
                c += 1
            if not cmp(d, k[lo + c - 1]) < 0:
                break
            k[lo + i - 1] = k[lo + c - 1]
            i = c
        k[lo + i - 1] = d

    def intro(lo, n, depth):
        while n > 1:
            if n <= 16:
                if n == 2:
                    sig(lo, 0, 1); return
                if n == 3:
                    sig(lo, 0, 1); sig(lo, 0, 2); sig(lo, 1, 2); return
                for i in range(n - 1):
                    t = k[lo + i + 1]; j = i
                    while j >= 0 and cmp(t, k[lo + j]) < 0:
                        k[lo + j + 1] = k[lo + j]; j -= 1
                    k[lo + j + 1] = t
                return
            if depth == 0:
                for i in range(n >> 1, 0, -1):
                    down(lo, i, n)
                for i in range(n, 1, -1):
                    k[lo], k[lo + i - 1] = k[lo + i - 1], k[lo]
                    down(lo, 1, i - 1)
                return
            depth -= 1
            hi = n - 1; mid = hi >> 1
            sig(lo, 0, mid); sig(lo, 0, hi); sig(lo, mid, hi)
            pivot = k[lo + mid]
            k[lo + mid], k[lo + hi - 1] = k[lo + hi - 1], k[lo + mid]
            left, right = 0, hi - 1
            while left < right:
                left += 1
                while cmp(k[lo + left], pivot) < 0:
                    left += 1
                right -= 1
                while cmp(pivot, k[lo + right]) < 0:
                    right -= 1
                if left >= right:
                    break
                k[lo + left], k[lo + right] = k[lo + right], k[lo + left]
            if left != hi - 1:
                k[lo + left], k[lo + hi - 1] = k[lo + hi - 1], k[lo + left]
            intro(lo + left + 1, n - (left + 1), depth)
            n = left
    n = len(k)
    if n >= 2:
        intro(0, n, 2 * (int(math.floor(math.log2(n))) + 1))
    return k


@pytest.mark.parametrize("n,ties", [(1, False), (2, False), (3, True), (16, True), (17, False), (100, True), (1000, False), (1500, True)])
def test_dotnet_introsort_restatement(orc, n, ties):
    rng = np.random.default_rng(n)
    key = (rng.integers(0, 7, n) if ties else rng.standard_normal(n)).astype(np.float32)
    idx = np.arange(n, dtype=np.int32)
    rng.shuffle(idx)
    want = _py_introsort(idx.tolist(), key)
    got = idx.copy()
    orc.lib().orc_dotnet_sort_by_key(got.ctypes.data, n, key.ctypes.data)
    assert sorted(got.tolist()) == list(range(n))
    assert np.all(np.diff(key[got]) >= 0)
    assert got.tolist() == want          # same permutation, ties included


def test_dotnet_introsort_heapsort_branch(orc):
    """Adversarial 'median-of-3 killer' input exhausts the depth limit -> heapsort path."""
    n = 4096
    key = np.zeros(n, np.float32)
    k = n // 2
    for i in range(1, k + 1):
        if i % 2 == 1:
            key[i - 1] = i
            key[i] = k + i
        key[k + i - 1] = 2 * i
    idx = np.arange(n, dtype=np.int32)
    got = idx.copy()
    orc.lib().orc_dotnet_sort_by_key(got.ctypes.data, n, key.ctypes.data)
    assert got.tolist() == _py_introsort(idx.tolist(), key)
    assert np.all(np.diff(key[got]) >= 0)


def test_config1_closed_form(orc):
    """Centre ray of config 1 hits the unit-diameter sphere at the closed-form distance; miss pixels
    carry PackRGBA8(sky(dir))."""
    cfg = scenes.CONFIGS[1]
    w = h = 65    # odd: pixel (32,32) is the exact image centre ray
    arrs, st, p = H.oracle_frame(orc, scenes.build_config1, cfg, w, h, 1)
    c = (h // 2) * w + w // 2
    assert arrs["gb_hitMask"][c] == 1
    o = np.array(cfg.cam_origin, np.float64); tgt = np.array(cfg.cam_lookat, np.float64)
    dist = np.linalg.norm(tgt - o) - 0.5
    assert abs(arrs["depth"][c] - dist) < 2e-6 * dist
    n = arrs["gb_normalWS"][c].astype(np.float64)
    assert np.allclose(n, (o - tgt) / np.linalg.norm(o - tgt), atol=2e-6)
    assert arrs["gb_objId"][c] == -1 and arrs["objectId"][c] == -1
    assert arrs["gb_matId"][c] == (0 | (1000 << 16))           # lambert, ior 1.0 -> 1000
    # a corner pixel misses: colour = sky(dir), depth = |dir*1e6|
    assert arrs["gb_hitMask"][0] == 0
    wp = arrs["gb_worldPos"][0].astype(np.float64)
    d = (wp - o) / np.linalg.norm(wp - o)
    tb = 0.5 * (d[1] + 1.0)
    sky = np.array([1, 1, 1]) * (1 - tb) + np.array([0.5, 0.7, 1.0]) * tb
    assert np.allclose(arrs["radiance"][0], sky, atol=1e-6)
    exp = 0xFF000000 | (int(255.99 * min(1, sky[0])) << 16) | (int(255.99 * min(1, sky[1])) << 8) | int(255.99 * min(1, sky[2]))
    assert int(arrs["color"][0]) & 0xFFFFFFFF == exp
    assert abs(arrs["depth"][0] - 1e6) < 1.0
    assert list(arrs["gb_normalWS"][0]) == [0, 1, 0] and arrs["gb_matId"][0] == -1
    # ray accounting: one primary per pixel; per hit pixel 1 shadow + 1 bounce (convex object, bounce escapes)
    hits = int(arrs["gb_hitMask"].sum())
    assert st.k[0].rays_closest == w * h and st.k[1].rays_closest == hits and st.k[1].diffuse_vertices == hits


def _random_rays(rng, n, origin_box, target_box):
    o = rng.uniform(origin_box[0], origin_box[1], (n, 3)).astype(np.float32)
    t = rng.uniform(target_box[0], target_box[1], (n, 3)).astype(np.float32)
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def test_bvh_equals_brute_force_spheres(orc):
    s = orc.OrcScene()
    scenes.build_random_spheres(s, 600, extent=6.0)
    o, d = _random_rays(np.random.default_rng(5), 4000, ((-8, 0.5, -8), (8, 6, 8)), ((-6, 0, -6), (6, 2, 6)))
    a = orc.trace_rays(s.desc(), o, d, brute=False)
    b = orc.trace_rays(s.desc(), o, d, brute=True)
    assert np.array_equal(a["hit"], b["hit"]) and a["hit"].mean() > 0.5
    assert np.array_equal(a["t"], b["t"])
    assert np.array_equal(a["shade"], b["shade"])
    assert np.array_equal(a["normal"], b["normal"])


def test_bvh_equals_brute_force_triangles(orc):
    s = orc.OrcScene()
    scenes.build_config4(s, 40, 40)
    o, d = _random_rays(np.random.default_rng(6), 4000, ((-3, 0.2, -3), (3, 3, 3)), ((-0.8, 0.4, -0.8), (0.8, 1.8, 0.8)))
    a = orc.trace_rays(s.desc(), o, d, brute=False)
    b = orc.trace_rays(s.desc(), o, d, brute=True)
    assert np.array_equal(a["hit"], b["hit"]) and a["hit"].mean() > 0.5
    assert np.array_equal(a["t"], b["t"])
    same_prim = a["objId"] == b["objId"]
    assert same_prim.mean() > 0.999          # exact-t ties on shared edges may name the other triangle


def test_multi_sphere_blas_bug_compatible(orc):
    """>4 spheres in ONE instance: the reference unions pre-computed bounds by array POSITION after
    the sort permuted idx[] (Scene.cs:386-395,413-419,452-455).  The restatement keeps the
    quirk: root bounds are right, child bounds are unions over positions."""
    s = orc.OrcScene()
    xs = [3.0, -2.0, 0.5, -4.0, 2.0, 1.0, -1.0, 4.0, -3.0]      # unsorted along x
    ids = [s.add_sphere(scenes.sphere((x, 0.0, 0.0), 0.25, (0.5, 0.5, 0.5))) for x in xs]
    s.build_sphere_instance(ids)
    s.rebuild_tlas()
    A = s.arrays()
    nodes = A["blasNodes"]
    assert nodes[0]["boundsMin"]["X"] == np.float32(-4.25) and nodes[0]["boundsMax"]["X"] == np.float32(4.25)
    right = nodes[1]                   # covers POSITIONS 4..8 -> spheres xs[4:9], not the 5 largest x
    assert right["boundsMin"]["X"] == np.float32(min(xs[4:]) - 0.25)
    assert right["boundsMax"]["X"] == np.float32(max(xs[4:]) + 0.25)
    assert len(A["spherePrimIdx"]) == 2 * len(xs)               # leaves append to the END of the list (:439-440)


def test_default_scene_layout(orc):
    """BuildDefaultScene: 6 spheres, each its own 1-node BLAS; prim index list = ids + leaf appends."""
    s = orc.OrcScene()
    s.build_default_scene()
    A = s.arrays()
    assert len(A["spheres"]) == 6 and len(A["instances"]) == 6 and len(A["blasNodes"]) == 6
    assert A["spherePrimIdx"].tolist() == [0, 1, 2, 3, 4, 5] * 2
    assert [int(n["first"]) for n in A["blasNodes"]] == [6, 7, 8, 9, 10, 11]
    assert len(A["texels"]) == 2 * 256 * 256 and len(A["texInfos"]) == 2
    assert len(A["tlasNodes"]) == 7                              # 6 instances, leaf <= 2: 3 leaves + 4 inner... checked below
    leaves = [n for n in A["tlasNodes"] if n["count"] > 0]
    assert sum(int(n["count"]) for n in leaves) == 6
    assert sorted(A["tlasInstanceIndices"].tolist()) == list(range(6))
    # skip pointers: memory order is [node][right subtree][left subtree]; left child's skip = right root
    root = A["tlasNodes"][0]
    assert root["right"] == 1 and A["tlasNodes"][int(root["left"])]["skipIndex"] == 1 and root["skipIndex"] == -1


def test_tlas_node_count_config3_formula(orc):
    """count>>1 median split with leaf <= 2: N(n) = 1 if n <= 2 else 1 + N(n - n//2) + N(n//2)."""
    def N(n):
        return 1 if n <= 2 else 1 + N(n - n // 2) + N(n // 2)
    s = orc.OrcScene()
    scenes.build_random_spheres(s, 200)
    A = s.arrays()
    assert len(A["tlasNodes"]) == N(201)
    assert N(10001) == 11809


def test_golden_fixtures(orc):
    """Oracle output on the committed small scenes equals the committed golden vectors
    (tests/golden/*.npz, produced by tests/golden/make_golden.py with this same oracle)."""
    from tests.golden import make_golden as G
    for name in G.CASES:
        path = os.path.join(GOLDEN, name + ".npz")
        assert os.path.exists(path), "missing fixture %s (run python -m tests.golden.make_golden)" % path
        want = np.load(path)
        got, st = G.render_case(orc, name)
        for k in got:
            assert np.all(H.bits_equal(want[k], got[k])), (name, k)
        meta = json.loads(str(want["counters_json"]))
        assert meta == [st.k[0].as_dict(), st.k[1].as_dict()]


def test_box_test_is_monotone_in_the_box(orc):
    """The property TracerFlat and the walker rest on: in IntersectAABB (SceneDeviceViews.cs:496-514)
    enlarging the box, or raising tMax, never turns a hit into a miss -- so a leaf whose own test passes has passed the
    tests of all its ancestors (their boxes are unions of their children's), and inner nodes only accelerate.
    Adversarial cases: axis-parallel rays (1/d replaced by 1e8), origins exactly on box planes, flat boxes, boxes that
    share planes with their parent, huge and tiny magnitudes.  Holds for finite slab arithmetic; rays whose 1/d is not
    finite are routed to the tree walk by the kernels."""
    rng = np.random.RandomState(11)
    n = 400000
    grid = np.array([-3.0, -1.0, -0.5, 0.0, 0.25, 0.5, 1.0, 2.0, 7.0], np.float32)

    def pick(shape, p_grid):
        v = (rng.rand(*shape).astype(np.float32) * 8 - 4)
        g = grid[rng.randint(0, len(grid), shape)]
        return np.where(rng.rand(*shape) < p_grid, g, v).astype(np.float32)

    a, b = pick((n, 3), 0.5), pick((n, 3), 0.5)
    clo, chi = np.minimum(a, b), np.maximum(a, b)
    flat = rng.rand(n, 3) < 0.15
    chi = np.where(flat, clo, chi)                                         # flat child boxes
    grow_lo = np.where(rng.rand(n, 3) < 0.4, 0, np.abs(pick((n, 3), 0.5))).astype(np.float32)
    grow_hi = np.where(rng.rand(n, 3) < 0.4, 0, np.abs(pick((n, 3), 0.5))).astype(np.float32)
    plo, phi = (clo - grow_lo).astype(np.float32), (chi + grow_hi).astype(np.float32)   # parent shares planes 40 % of the time
    o = pick((n, 3), 0.6)
    on_plane = rng.rand(n, 3) < 0.2
    o = np.where(on_plane, np.where(rng.rand(n, 3) < 0.5, clo, phi), o).astype(np.float32)   # origin exactly on a plane
    d = pick((n, 3), 0.3)
    d = np.where(rng.rand(n, 3) < 0.25, 0.0, d).astype(np.float32)         # axis-parallel components
    d = np.where(rng.rand(n, 3) < 0.05, d * np.float32(1e-30), d).astype(np.float32)
    scale = np.float32(10.0) ** rng.randint(-3, 4, (n, 1)).astype(np.float32)
    clo, chi, plo, phi, o = [(x * scale).astype(np.float32) for x in (clo, chi, plo, phi, o)]
    tmax_c = np.where(rng.rand(n) < 0.5, 1e30, rng.rand(n) * 20).astype(np.float32)
    tmax_p = np.maximum(tmax_c, np.where(rng.rand(n) < 0.5, tmax_c, 1e30).astype(np.float32))
    inv = 1.0 / np.where(d != 0, d, np.float32(1e-8)).astype(np.float32)
    ok = np.isfinite(inv.astype(np.float32)).all(axis=1)
    hc = orc.hit_box(o, d, clo, chi, tmax_c)
    hp = orc.hit_box(o, d, plo, phi, tmax_p)
    bad = np.nonzero((hc == 1) & (hp == 0) & ok)[0]
    assert len(bad) == 0, (o[bad[0]], d[bad[0]], clo[bad[0]], chi[bad[0]], plo[bad[0]], phi[bad[0]], tmax_c[bad[0]], tmax_p[bad[0]])
    assert hc.sum() > n // 50 and (hp.sum() - hc.sum()) > n // 100        # the sample exercises both outcomes
    # why the kernels guard on finite 1/d: with a denormal direction component 1/d = inf and 0 * inf = NaN drops out of min/max
    o1 = np.array([[1.0, 0.2, 0.3]], np.float32); d1 = np.array([[1e-40, 0.5, 0.8]], np.float32)
    child = orc.hit_box(o1, d1, [[1.0, 0.0, 0.0]], [[1.0, 5.0, 5.0]], [1e30])[0]
    parent = orc.hit_box(o1, d1, [[1.0, 0.0, 0.0]], [[2.0, 5.0, 5.0]], [1e30])[0]
    assert (child, parent) == (1, 0)


def test_xorshift_step_is_a_bijection_and_keeps_zero_out():
    """RTUtils.cs:33-42 guards the xorshift state with `x != 0 ? x : 1`.  The kernels drop the guard (hrt_device.hpp, Rng::next_u):
    the step x ^= x << 13; x ^= x >> 17; x ^= x << 5 is a linear map of GF(2)^32 of full rank, so only a zero state yields zero, and
    no state is ever zero (a zero seed becomes 1, the seed mixer ORs in 1)."""
    def step(x):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        return x
    rows = [step(1 << i) for i in range(32)]            # images of the basis vectors
    rank = 0
    for bit in range(32):                                # Gaussian elimination over GF(2)
        piv = next((i for i in range(rank, 32) if rows[i] >> bit & 1), None)
        if piv is None:
            continue
        rows[rank], rows[piv] = rows[piv], rows[rank]
        for i in range(32):
            if i != rank and rows[i] >> bit & 1:
                rows[i] ^= rows[rank]
        rank += 1
    assert rank == 32
    rng = np.random.default_rng(5)
    for x in [1, 0xFFFFFFFF, 0x80000000] + [int(v) for v in rng.integers(1, 2 ** 32, 2000)]:
        a, b = int(rng.integers(1, 2 ** 32)), int(rng.integers(1, 2 ** 32))
        assert step(x) != 0 and step(a ^ b) == step(a) ^ step(b)      # non-zero image, linearity
