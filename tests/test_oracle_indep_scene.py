"""The scene BUILDERS restated a second time (oracle/orc_indep_scene.py, plain Python written from Scene.cs alone) against the
oracle's C++ restatement (oracle/orc_scene.hpp) and the product's host builder (csrc/hrt_host.cpp): all three must emit the same
fifteen arrays.  Two independent readings agreeing is the only pin the builders have (the reference holds no fixture for them):
it found the position-indexed triangle lookup of later meshes and the host flavour of Min / Max (.NET Math.Min / Max return a NaN
operand; the kernels' min / max return the other one).  NaN payloads are not compared (any NaN equals any NaN, as everywhere)."""
import numpy as np
import pytest

from ilgpu_raytracing_amd import engine, scenes
from oracle.orc_indep_scene import IndepScene
from tests import test_fuzz_gpu as FZ, test_hostile_gpu as HO


def _fields_equal(a, b, path):
    if a.dtype.names:
        for f in a.dtype.names:
            _fields_equal(a[f], b[f], path + "." + f)
        return
    if a.dtype.kind == "f":
        ua, ub = a.view(np.uint32), b.view(np.uint32)
        same = (ua == ub) | (np.isnan(a) & np.isnan(b))
    else:
        same = a == b
    assert bool(np.all(same)), "%s differs at %s" % (path, np.flatnonzero(~np.asarray(same).reshape(-1))[:8])


def _same(ref, got, who):
    A, B = ref.arrays(), got.arrays()
    assert set(A) == set(B)
    for k in A:
        assert A[k].dtype == B[k].dtype and A[k].shape == B[k].shape, (who, k, A[k].shape, B[k].shape)
        _fields_equal(A[k], B[k], who + ":" + k)


def _three(orc, build):
    a, b, c = orc.OrcScene(), IndepScene(), engine.Scene()
    for s in (a, b, c):
        build(s)
    _same(a, b, "python restatement")
    _same(a, c, "host builder")


def _with_tlas(build):
    def f(b):
        build(b)
        b.rebuild_tlas()
    return f


NAMED = {
    "config1": scenes.build_config1,
    "config2": scenes.build_config2,
    "textured": scenes.build_textured_test_scene,
    "random_spheres_300": lambda b: scenes.build_random_spheres(b, 300),
    "blob_20x20": lambda b: scenes.build_config4(b, 20, 20),
}


@pytest.mark.parametrize("name", list(NAMED))
def test_named_scenes(orc, hrt_lib, name):
    _three(orc, NAMED[name])


HOSTILE = {
    "degenerate_spheres": HO._degenerate_spheres(),
    "nonfinite_spheres": HO._nonfinite_spheres(),             # NaN / inf centres and radii: NaN boxes all the way up the TLAS
    "degenerate_mesh": _with_tlas(HO._degenerate_mesh),       # a NaN vertex
    "odd_transforms": _with_tlas(HO._odd_transforms),         # a NaN matrix entry, zero / negative / huge scale
    "odd_textures": _with_tlas(HO._odd_textures),             # four meshes: the position-indexed lookup
}


@pytest.mark.parametrize("name", list(HOSTILE))
def test_hostile_scenes(orc, hrt_lib, name):
    _three(orc, HOSTILE[name])


@pytest.mark.parametrize("case", range(24))
def test_random_recipes(orc, hrt_lib, case):
    ops, _ = FZ._scene_recipe(0x1D5C0000 + case)
    _three(orc, lambda b: FZ._apply(b, ops))


def test_host_min_max_known_answers(orc):
    """Math.Min / Max of .NET on the operand pairs where they differ from minNum / maxNum, through a builder: the box of a
    one-sphere BLAS with a NaN centre coordinate is NaN in that coordinate (minNum would leave float.MaxValue there)."""
    a = orc.OrcScene()
    a.build_sphere_instance([a.add_sphere(scenes.sphere((float("nan"), 1.0, 2.0), 0.5, (0.5, 0.5, 0.5)))])
    a.rebuild_tlas()
    A = a.arrays()
    n = A["blasNodes"][0]
    assert np.isnan(n["boundsMin"]["X"]) and np.isnan(n["boundsMax"]["X"])
    assert n["boundsMin"]["Y"] == np.float32(0.5) and n["boundsMax"]["Z"] == np.float32(2.5)
    # TransformAABB multiplies every corner through the matrix: 0 * NaN reaches all three rows, and the TLAS root unites that
    for box in (A["instances"][0]["worldBoundsMin"], A["instances"][0]["worldBoundsMax"], A["tlasNodes"][0]["boundsMin"], A["tlasNodes"][0]["boundsMax"]):
        assert all(np.isnan(box[f]) for f in "XYZ")


# ------------------------------------------------------------------ camera / sun: Camera.cs and RTRenderer.cs host code, three ways
def _cam_vec(c):
    if isinstance(c, dict):
        out = [x for k in ("origin", "lowerLeft", "horizontal", "vertical", "forward", "right", "up") for x in c[k]] + [c["aspect"], c["fovYRadians"]]
    else:
        out = [getattr(getattr(c, k), f) for k in ("origin", "lowerLeft", "horizontal", "vertical", "forward", "right", "up") for f in "XYZ"] + [c.aspect, c.fovYRadians]
    return np.array(out, np.float32)


def _same_bits(a, b):
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_camera_and_sun_three_ways(orc, hrt_lib):
    """The look-at constructor (Camera.cs:100-126), BakeCameraDerived (RTRenderer.cs:241-263) and the sun direction (:174-178) from
    the oracle's C++, the independent Python and the product's host code, on 400 random parameter sets of which a third are
    degenerate (a camera looking at itself, straight up, along its up hint; NaN / infinite / zero / huge positions, fields of view,
    aspects; frame vectors too short for the bake's divisions)."""
    from oracle import orc_indep_scene as I

    def math(name, x):
        return orc.math_eval(name, np.array([x], np.float32), None)[0]
    rng = np.random.default_rng(77)
    nan, inf = float("nan"), float("inf")
    pool = [nan, inf, -inf, 0.0, 1e30, -1e30, 1e-30]
    for case in range(400):
        o = [float(v) for v in rng.uniform(-5, 5, 3)]; l = [float(v) for v in rng.uniform(-2, 2, 3)]; up = [0.0, 1.0, 0.0]
        vf = float(rng.choice([35.0, 60.0, 90.0, float(rng.uniform(1, 179))])); asp = float(rng.uniform(0.3, 3.0)); foc = float(rng.choice([1.0, 2.5]))
        k = rng.random()
        if k < 0.08: l = list(o)
        elif k < 0.16: l = [o[0], o[1] + 1.0, o[2]]
        elif k < 0.2: up = [1.0, 0.0, 0.0]; l = [o[0] + 1.0, o[1], o[2]]
        elif k < 0.36:
            t, v = int(rng.integers(0, 5)), float(rng.choice(pool))
            if t == 0: o[int(rng.integers(0, 3))] = v
            elif t == 1: l[int(rng.integers(0, 3))] = v
            elif t == 2: vf = float(rng.choice([0.0, 180.0, 360.0, -60.0, nan, inf]))
            elif t == 3: asp = v
            else: up = [v, 1.0, 0.0]
        a, b, c = orc.camera_lookat(o, l, up, vf, asp, foc), I.camera_lookat(math, o, l, up, vf, asp, foc), engine.camera_look_at(o, l, up, vf, asp, foc)
        assert _same_bits(_cam_vec(a), _cam_vec(b)) and _same_bits(_cam_vec(a), _cam_vec(c)), ("look-at", case, o, l, up, vf, asp)
        if rng.random() < 0.2:              # frame vectors below the bake's 1e-6 thresholds: its fallbacks run
            for cam in (a, c):
                for k_ in ("horizontal", "vertical"):
                    for f_ in "XYZ":
                        setattr(getattr(cam, k_), f_, float(np.float32(getattr(getattr(cam, k_), f_)) * np.float32(1e-8)))
            b["horizontal"] = tuple(np.float32(getattr(a.horizontal, f_)) for f_ in "XYZ")
            b["vertical"] = tuple(np.float32(getattr(a.vertical, f_)) for f_ in "XYZ")
        w, h = int(rng.integers(1, 200)), int(rng.integers(0, 200))
        orc.camera_bake(a, w, h); b = I.camera_bake(math, b, w, h); engine.bake_camera_derived(c, w, h)
        assert _same_bits(_cam_vec(a), _cam_vec(b)) and _same_bits(_cam_vec(a), _cam_vec(c)), ("bake", case)
        az, el = float(rng.uniform(-7, 7)), float(rng.choice([float(rng.uniform(-2, 2)), nan, inf, 0.0, 1.5707964]))
        s1, s2, s3 = (np.array(v, np.float32) for v in (orc.sun_dir(az, el), I.sun_dir(math, az, el), engine.sun_direction(az, el)))
        assert _same_bits(s1, s2) and _same_bits(s1, s3), ("sun", az, el)
