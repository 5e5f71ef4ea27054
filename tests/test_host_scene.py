"""Host side above the C ABI (csrc/hrt_host.cpp: scene builders, camera) against the oracle's literal
restatement of Engine/Scene.cs / Engine/Camera.cs: every emitted array must be byte-identical.
Runs without a GPU (the host code is plain C++ inside libhip_raytrace.so)."""
import ctypes as C

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes


def _same_arrays(a, b):
    A, B = a.arrays(), b.arrays()
    for k in A:
        assert A[k].dtype == B[k].dtype and len(A[k]) == len(B[k]), k
        assert A[k].tobytes() == B[k].tobytes(), "array %s differs between host builder and oracle" % k
    return A


BUILDERS = {
    "config1": scenes.build_config1,
    "config2": scenes.build_config2,
    "default_scene": lambda b: b.build_default_scene(),
    "random_spheres_777": lambda b: scenes.build_random_spheres(b, 777),
    "textured": scenes.build_textured_test_scene,
    "blob_40x40": lambda b: scenes.build_config4(b, 40, 40),
    "blob_97x61": lambda b: scenes.build_config4(b, 97, 61),          # odd counts: ragged median splits
    "terrain_120": lambda b: scenes.build_config5(b, 120),            # > 8192 tris: threaded subtree build
}


@pytest.mark.parametrize("name", list(BUILDERS))
def test_builder_matches_oracle(orc, hrt_lib, name):
    a, b = orc.OrcScene(), engine.Scene()
    BUILDERS[name](a)
    BUILDERS[name](b)
    A = _same_arrays(a, b)
    # structural sanity of the skip-pointer layout the kernels rely on
    for nodes in (A["tlasNodes"], A["blasNodes"]):
        for i, n in enumerate(nodes):
            if n["count"] == 0 and n["left"] != -1:
                assert n["right"] == i + 1 and n["left"] > n["right"]


def test_multi_sphere_instance_bug_compatible(orc, hrt_lib):
    """9 spheres in one instance (reference quirk F4: bounds by array position) -- identical arrays."""
    def build(b):
        ids = [b.add_sphere(scenes.sphere((x, 0.1 * i, -0.2 * i), 0.25, (0.5, 0.5, 0.5)))
               for i, x in enumerate([3.0, -2.0, 0.5, -4.0, 2.0, 1.0, -1.0, 4.0, -3.0])]
        b.build_sphere_instance(ids)
        b.rebuild_tlas()
    a, b = orc.OrcScene(), engine.Scene()
    build(a); build(b)
    _same_arrays(a, b)


def test_many_tied_centroids(orc, hrt_lib):
    """A flat regular grid has thousands of equal centroid keys: topology then depends on the
    (unstable) .NET introsort restatement; host builder and oracle must agree on it."""
    def build(b):
        n = 30
        s = np.linspace(-1, 1, n + 1)
        zz, xx = np.meshgrid(s, s, indexing="ij")
        b.load_mesh_instance(scenes.grid_mesh(xx, np.zeros_like(xx), zz, xx, zz))
    a, b = orc.OrcScene(), engine.Scene()
    build(a); build(b)
    _same_arrays(a, b)


def test_several_meshes_position_indexed_quirk(orc, hrt_lib):
    """BuildBLAS_Triangles looks the triangles of a mesh up BY POSITION in the prim-index list (Scene.cs:398-403, :428), where the
    leaf copies of every earlier mesh sit in between: the BLAS of the second, third, ... mesh of a scene is built over the triangles
    that window happens to name, not its own.  Both builders restate that literally; the known answer below pins which triangles
    those are for three 2x2-quad meshes (8 triangles each)."""
    def build(b):
        s = np.linspace(-1.0, 1.0, 3)
        yq, xq = np.meshgrid(s, s, indexing="ij")
        for k in range(3):
            b.load_mesh_instance(scenes.grid_mesh(xq + 3.0 * k, yq, 0.1 * xq * yq - k, (xq + 1) / 2, (yq + 1) / 2))
    a, b = orc.OrcScene(), engine.Scene()
    build(a); build(b)
    A = _same_arrays(a, b)
    prim = [int(v) for v in A["triPrimIdx"]]
    assert len(prim) == 3 * 16                               # per mesh: 8 identity entries, then 8 leaf copies
    assert prim[0:8] == list(range(8)) and prim[16:24] == list(range(8, 16)) and prim[32:40] == list(range(16, 24))
    assert sorted(prim[8:16]) == list(range(8))              # mesh 0: its own triangles
    assert sorted(prim[24:32]) == list(range(8))             # mesh 1: window [8, 16) of the list = mesh 0's leaf copies
    assert sorted(prim[40:48]) == list(range(8, 16))         # mesh 2: window [16, 24) = mesh 1's identity entries
    inst = A["instances"]
    assert [int(r["primIndexFirst"]) for r in inst] == [0, 8, 16]


def test_empty_scene(orc, hrt_lib):
    a, b = orc.OrcScene(), engine.Scene()
    a.rebuild_tlas(); b.rebuild_tlas()
    A = _same_arrays(a, b)
    assert len(A["tlasNodes"]) == 1 and A["tlasNodes"][0]["count"] == 0 and A["tlasNodes"][0]["left"] == -1


def test_invalid_arguments_rejected(hrt_lib):
    s = engine.Scene()
    with pytest.raises(ValueError):
        s.build_sphere_instance([3])                       # no such sphere
    sid = s.add_sphere(scenes.sphere((0, 0, 0), 1, (1, 1, 1)))
    assert sid == 0
    m = scenes.grid_mesh(*[np.zeros((2, 2))] * 5)
    m.triangles[0, 0] = 99                                 # vertex index out of range
    with pytest.raises(ValueError):
        s.load_mesh_instance(m)


def _cam_bytes(c):
    return bytes(C.string_at(C.byref(c), C.sizeof(T.Camera)))


def test_camera_functions_match_oracle(orc, hrt_lib):
    for w, h, fov in [(1280, 720, 60.0), (256, 256, 45.0), (1920, 1080, 90.0)]:
        a, b = orc.camera_create(w, h, fov), engine.create_camera(w, h, fov)
        assert _cam_bytes(a) == _cam_bytes(b)
        orc.camera_translate(a, (1, 0, -4)); engine.camera_translate(b, (1, 0, -4))
        assert _cam_bytes(a) == _cam_bytes(b)
        orc.camera_bake(a, w, h); engine.bake_camera_derived(b, w, h)
        assert _cam_bytes(a) == _cam_bytes(b)
    rng = np.random.default_rng(2)
    for _ in range(50):
        o, l = rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 3)
        up = (0, 1, 0) if rng.random() < 0.8 else tuple(rng.standard_normal(3))
        asp = float(rng.uniform(0.5, 2.5)); fov = float(rng.uniform(20, 100))
        a, b = orc.camera_lookat(o, l, up, fov, asp), engine.camera_look_at(o, l, up, fov, asp)
        assert _cam_bytes(a) == _cam_bytes(b)
    # looking straight down: up hint is parallel to forward -> OrthoBasis fallback (Camera.cs:197-201)
    a, b = orc.camera_lookat((0, 5, 0), (0, 0, 0), (0, 1, 0), 60, 1.5), engine.camera_look_at((0, 5, 0), (0, 0, 0), (0, 1, 0), 60, 1.5)
    assert _cam_bytes(a) == _cam_bytes(b)
    for az, el in [(0.0, 0.9), (1.5707963, 0.6), (3.0, 0.1), (5.5, 1.4)]:
        assert orc.sun_dir(az, el) == engine.sun_direction(az, el)
    d = engine.sun_direction(0.0, 0.9)
    assert abs(d[0] - 0.6216) < 1e-4 and abs(d[1] - 0.7833) < 1e-4 and d[2] == 0.0


def test_default_renderer_camera(hrt_lib):
    """RTRenderer ctor camera: CreateCamera(w,h,60) then Translate((1,0,-4)) -> origin (1,1,-1)."""
    c = engine.create_camera(1280, 720, 60.0)
    engine.camera_translate(c, (1, 0, -4))
    assert (c.origin.X, c.origin.Y, c.origin.Z) == (1.0, 1.0, -1.0)
    assert abs(c.forward.Y + 0.164) < 1e-3 and abs(c.forward.Z + 0.986) < 1e-3


MOVES = [("x", 0.0, 1.0, (0.25, -0.5, 3.0)), ("y", 33.0, 1.0, (1.0, 0.0, -2.0)), ("z", -71.5, 0.6, (0.0, 0.0, 0.0)),
         ("x", 180.0, 2.5, (-4.0, 0.125, 0.75))]


@pytest.mark.parametrize("name", ["random_spheres_777", "textured", "blob_40x40"])
def test_moved_instances_match_oracle(orc, hrt_lib, name):
    """hrth_scene_set_instance_transform (the host mirror of hrt_scene_update_instances) == the oracle's restatement of
    what BuildSphereInstance / LoadObjInstance derive from objectToWorld (Scene.cs:395-402,236-252,560-580,616-638);
    RebuildTLAS afterwards gives the same tree on both sides."""
    a, b = orc.OrcScene(), engine.Scene()
    BUILDERS[name](a)
    BUILDERS[name](b)
    n = len(a.arrays()["instances"])
    for k in range(min(n, 9)):
        m = scenes.rotation_affine(*MOVES[k % len(MOVES)])
        a.set_instance_transform(n - 1 - k, m)
        b.set_instance_transform(n - 1 - k, m)
    _same_arrays(a, b)
    a.rebuild_tlas(); b.rebuild_tlas()
    _same_arrays(a, b)
    with pytest.raises(IndexError):
        b.set_instance_transform(n, T.identity_affine())


def test_moved_instance_known_answers(orc):
    """Closed forms: a translation keeps the matrix exact; a uniform scale s gives uniformScale = s and worldToObject =
    I / s (normalised columns times 1 / s, Scene.cs:625-631); a quarter turn about y maps
    the unit box onto itself with worldToObject equal to the ROTATION, not its transpose (the quirk DESIGN.md 4 records)."""
    s = orc.OrcScene()
    sid = s.add_sphere(scenes.sphere((0.0, 0.0, 0.0), 1.0, (1.0, 1.0, 1.0)))
    s.build_sphere_instance([sid])
    s.set_instance_transform(0, scenes.rotation_affine("y", 0.0, 1.0, (2.0, -3.0, 0.5)))
    i = s.arrays()["instances"][0]
    assert tuple(i["worldBoundsMin"].tolist()) == (1.0, -4.0, -0.5) and tuple(i["worldBoundsMax"].tolist()) == (3.0, -2.0, 1.5)
    w = i["worldToObject"]
    assert (w["m00"], w["m11"], w["m22"], w["m03"], w["m13"], w["m23"]) == (1.0, 1.0, 1.0, -2.0, 3.0, -0.5) and i["uniformScale"] == 1.0
    s.set_instance_transform(0, scenes.rotation_affine("y", 0.0, 4.0, (0.0, 0.0, 0.0)))
    i = s.arrays()["instances"][0]
    assert i["uniformScale"] == 4.0 and i["worldToObject"]["m00"] == 0.25 and tuple(i["worldBoundsMax"].tolist()) == (4.0, 4.0, 4.0)
    q = T.identity_affine()
    q.m00, q.m02, q.m20, q.m22 = 0.0, 1.0, -1.0, 0.0                   # x -> -z, z -> x
    s.set_instance_transform(0, q)
    i = s.arrays()["instances"][0]
    w = i["worldToObject"]
    assert (w["m00"], w["m02"], w["m20"], w["m22"]) == (0.0, 1.0, -1.0, 0.0)
    assert tuple(i["worldBoundsMin"].tolist()) == (-1.0, -1.0, -1.0) and tuple(i["worldBoundsMax"].tolist()) == (1.0, 1.0, 1.0)
