"""ctypes / numpy mirrors of include/hrt_types.h and include/hip_raytrace.h.

Every Structure here has the field order and size of the C struct of the same name
(which in turn follows the reference's C# struct; citations are in hrt_types.h).
`np_dtype(S)` gives the matching numpy structured dtype so scene arrays can be built
with numpy and handed to the C ABI without copies.
"""
import ctypes as C
import numpy as np


class Float3(C.Structure):
    _fields_ = [("X", C.c_float), ("Y", C.c_float), ("Z", C.c_float)]


class Float2(C.Structure):
    _fields_ = [("X", C.c_float), ("Y", C.c_float)]


class Affine3x4(C.Structure):
    _fields_ = [(f"m{r}{c}", C.c_float) for r in range(3) for c in range(4)]


class MaterialRecord(C.Structure):
    _fields_ = [("Kd", Float3), ("HasDiffuseMap", C.c_int32), ("DiffuseTexIndex", C.c_int32),
                ("Shading", C.c_int32), ("IOR", C.c_float), ("HasAlphaMap", C.c_int32),
                ("AlphaTexIndex", C.c_int32), ("TwoSided", C.c_int32), ("AlphaCutoff", C.c_float)]


class MeshDesc(C.Structure):          # hrth_mesh_desc (include/hrt_host.h)
    _fields_ = [("positions", C.c_void_p), ("n_positions", C.c_int), ("triangles", C.c_void_p), ("n_triangles", C.c_int),
                ("texcoords", C.c_void_p), ("n_texcoords", C.c_int), ("tri_uvs", C.c_void_p),
                ("tri_material_index", C.c_void_p), ("n_tri_material_index", C.c_int),
                ("materials", C.POINTER(MaterialRecord)), ("n_materials", C.c_int),
                ("tex_w", C.POINTER(C.c_int)), ("tex_h", C.POINTER(C.c_int)), ("tex_bgra", C.POINTER(C.c_uint8)), ("n_textures", C.c_int)]


HRTH_ERR_ARGUMENT, HRTH_ERR_NOT_FOUND, HRTH_ERR_FORMAT = -1, -2, -3


class Sphere(C.Structure):
    _fields_ = [("center", Float3), ("radius", C.c_float), ("albedo", Float3),
                ("material", MaterialRecord), ("shading", C.c_int32), ("ior", C.c_float)]


class BvhNode(C.Structure):
    _fields_ = [("boundsMin", Float3), ("boundsMax", Float3), ("left", C.c_int32), ("right", C.c_int32),
                ("first", C.c_int32), ("count", C.c_int32), ("skipIndex", C.c_int32)]


class InstanceRecord(C.Structure):
    _fields_ = [("type", C.c_int32), ("blasRoot", C.c_int32), ("blasNodeCount", C.c_int32),
                ("primIndexFirst", C.c_int32), ("primIndexCount", C.c_int32),
                ("objectToWorld", Affine3x4), ("worldToObject", Affine3x4), ("uniformScale", C.c_float),
                ("worldBoundsMin", Float3), ("worldBoundsMax", Float3)]


class MeshTri(C.Structure):
    _fields_ = [("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32)]


class MeshTriUV(C.Structure):
    _fields_ = [("t0", C.c_int32), ("t1", C.c_int32), ("t2", C.c_int32)]


class RGBA32(C.Structure):
    _fields_ = [("R", C.c_uint8), ("G", C.c_uint8), ("B", C.c_uint8), ("A", C.c_uint8)]


class TexInfo(C.Structure):
    _fields_ = [("Offset", C.c_int32), ("Width", C.c_int32), ("Height", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("origin", Float3), ("lowerLeft", Float3), ("horizontal", Float3), ("vertical", Float3),
                ("forward", Float3), ("right", Float3), ("up", Float3),
                ("aspect", C.c_float), ("fovYRadians", C.c_float)]


_SIZES = {Float3: 12, Float2: 8, Affine3x4: 48, MaterialRecord: 44, Sphere: 80, BvhNode: 44,
          InstanceRecord: 144, MeshTri: 12, MeshTriUV: 12, RGBA32: 4, TexInfo: 12, Camera: 92}
for _s, _n in _SIZES.items():
    assert C.sizeof(_s) == _n, (_s, C.sizeof(_s), _n)

# order = fields of SceneDeviceViews (SceneDeviceViews.cs:13-27)
SCENE_ARRAYS = [
    ("tlasNodes", BvhNode), ("tlasInstanceIndices", C.c_int32), ("instances", InstanceRecord),
    ("blasNodes", BvhNode), ("spherePrimIdx", C.c_int32), ("spheres", Sphere),
    ("triPrimIdx", C.c_int32), ("meshPositions", Float3), ("meshTris", MeshTri),
    ("meshTexcoords", Float2), ("meshTriUVs", MeshTriUV), ("triMatIndex", C.c_int32),
    ("materials", MaterialRecord), ("texels", RGBA32), ("texInfos", TexInfo),
]


REBUILD_AUTO, REBUILD_FORCE_REFIT, REBUILD_FORCE_REBUILD = 0, 1, 2      # hrt_rebuild_policy = RebuildPolicy (BvhManager.cs:13-18)
REBUILD_BLAS = 16                                                       # flag of hrt_scene_update_positions: new BLAS topology as well


class BvhUpdateStats(C.Structure):    # hrt_bvh_update_stats
    _fields_ = [("action", C.c_int32), ("tlas_nodes", C.c_int32), ("tlas_slots", C.c_int32), ("general_instances", C.c_int32),
                ("growth_refit", C.c_float), ("growth_final", C.c_float), ("sah_cost", C.c_float), ("device_ms", C.c_float),
                ("blas_action", C.c_int32), ("blas_growth", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [f for name, t in SCENE_ARRAYS for f in ((name, C.POINTER(t)), ("n_" + name, C.c_int64))]


class FrameParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("frame", C.c_int32),
                ("cam", Camera), ("prevCam", Camera),
                ("dirLightDir", Float3), ("dirLightRadiance", Float3),
                ("skyTintTop", Float3), ("skyTintBottom", Float3),
                ("debugCamSeq", C.c_int32), ("enableTemporalReuse", C.c_int32),
                ("enableSpatialReuse", C.c_int32), ("rngLockNoise", C.c_int32),
                ("spp", C.c_int32), ("maxDepth", C.c_int32)]


COUNTER_FIELDS = ["rays_closest", "rays_shadow", "node_visits", "leaf_instances", "sphere_tests",
                  "tri_tests", "tri_mt_hits", "tri_accepted", "reuse_imports", "diffuse_vertices"]


class KernelCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


class Stats(C.Structure):
    _fields_ = [("k", KernelCounters * 2), ("kernel_ms", C.c_double * 2), ("d2h_ms", C.c_double),
                ("n_devices", C.c_int32), ("counters_valid", C.c_int32), ("frames", C.c_int32), ("reserved", C.c_int32)]


OUTPUT_ARRAYS = [
    ("color", np.int32, 1), ("depth", np.float32, 1), ("objectId", np.int32, 1), ("cameraId", np.int32, 1),
    ("radiance", np.float32, 3),
    ("gb_worldPos", np.float32, 3), ("gb_normalWS", np.float32, 3), ("gb_baseColor", np.float32, 3),
    ("gb_matId", np.int32, 1), ("gb_objId", np.int32, 1), ("gb_hitMask", np.int32, 1),
    ("res_L", np.float32, 3), ("res_wi", np.float32, 3), ("res_pdf", np.float32, 1), ("res_w", np.float32, 1),
    ("res_wSum", np.float32, 1), ("res_m", np.int32, 1), ("res_lightId", np.int32, 1),
]


class Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _, _ in OUTPUT_ARRAYS]


class RenderOpts(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("row_begin", C.c_int32), ("row_end", C.c_int32), ("strip_n", C.c_int32), ("strip_i", C.c_int32)]


class DeviceViews(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("row_begin", "row_end", "strip_n", "strip_i", "width", "height", "device_id", "reserved")] + \
               [(n, C.c_void_p) for n in ("color", "depth", "objectId", "radiance", "gb_worldPos", "gb_normalWS",
                                          "gb_baseColor", "gb_matId", "gb_objId", "gb_hitMask", "present_color")] + \
               [("present_width", C.c_int32), ("present_height", C.c_int32), ("res_a", C.c_void_p * 7), ("res_b", C.c_void_p * 7)]


class PresentParams(C.Structure):
    _fields_ = [("out_width", C.c_int32), ("out_height", C.c_int32), ("mode", C.c_int32),
                ("feedback", C.c_float), ("sharpness", C.c_float), ("clampK", C.c_float)]


PRESENT_RESAMPLE, PRESENT_TAAU = 0, 1


FLAG_COUNTERS = 1
FLAG_SKIP_PRIMARY = 2
FLAG_REFERENCE_LAYOUT = 4
FLAG_NO_SYNC = 8
FLAG_MEGAKERNEL = 16
FLAG_STREAMED = 32
FLAG_PRIMARY_ONLY = 64
FLAG_EXCHANGED = 128
FLAG_TREELETS = 256

SHADING_LAMBERT, SHADING_MIRROR, SHADING_GLASS = 0, 1, 2
BLAS_SPHERESET, BLAS_TRIMESH = 1, 2


def np_dtype(struct):
    """numpy structured dtype with the exact layout of a ctypes Structure (or scalar)."""
    return np.dtype(struct)


def f3(x, y, z):
    return Float3(float(x), float(y), float(z))


def identity_affine():
    a = Affine3x4()
    a.m00 = a.m11 = a.m22 = 1.0
    return a


def alloc_outputs(width, height, names=None):
    """dict of numpy arrays for the requested output names + the Outputs struct that points at them."""
    P = width * height
    arrs, o = {}, Outputs()
    for n, dt, k in OUTPUT_ARRAYS:
        if names is not None and n not in names:
            continue
        cnt = 1 if n == "cameraId" else P
        a = np.zeros((cnt, k) if k > 1 else (cnt,), dtype=dt)
        arrs[n] = a
        setattr(o, n, a.ctypes.data)
    return arrs, o


def scene_desc_from_arrays(arrays):
    """arrays: dict name -> numpy array (structured or int32).  Returns (SceneDesc, keepalive)."""
    d = SceneDesc()
    keep = []
    for name, t in SCENE_ARRAYS:
        a = arrays.get(name)
        if a is None or len(a) == 0:
            setattr(d, "n_" + name, 0)
            continue
        a = np.ascontiguousarray(a)
        assert a.dtype.itemsize == C.sizeof(t), (name, a.dtype.itemsize, C.sizeof(t))
        keep.append(a)
        setattr(d, name, C.cast(a.ctypes.data, C.POINTER(t)))
        setattr(d, "n_" + name, len(a))
    return d, keep


def arrays_from_scene_desc(d):
    """Copy the 15 arrays a SceneDesc points at into numpy arrays (dict name -> array)."""
    out = {}
    for name, t in SCENE_ARRAYS:
        n = getattr(d, "n_" + name)
        dt = np_dtype(t)
        if n == 0:
            out[name] = np.zeros(0, dtype=dt)
            continue
        p = getattr(d, name)
        buf = (C.c_char * (n * C.sizeof(t))).from_address(C.addressof(p.contents))
        out[name] = np.frombuffer(buf, dtype=dt, count=n).copy()
    return out
