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
