"""oracle/orc.py -- ctypes binding of oracle/liborc.so.  TEST INFRASTRUCTURE ONLY.

Importers: tests/, __graft_entry__.smoke(), bench.py (cpu_baseline leg).  The package
ilgpu_raytracing_amd never imports this module (tests/test_layout.py checks that).
"""
import ctypes as C
import os
import subprocess
import numpy as np

from ilgpu_raytracing_amd import _types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_VARIANT = ""          # "" = liborc.so; "dotnet" = liborc_dotnet.so (kernel min / max by the CPUAccelerator rule, include/hrt_math.h)
_LIBS = {}


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_api.cpp", "orc_kernels.hpp", "orc_scene.hpp", "orc_post.hpp")] + \
           [os.path.join(_HERE, "..", "include", f) for f in ("hrt_types.h", "hrt_math.h", "hip_raytrace.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def set_variant(name=""):
    """Switches every function of this module to another build of the oracle ("" = the contract's; "dotnet")."""
    global _LIB, _VARIANT
    _LIBS[_VARIANT] = _LIB
    _VARIANT = name
    _LIB = _LIBS.get(name)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc%s.so" % ("_" + _VARIANT if _VARIANT else ""))
        if not os.path.exists(so) or (_VARIANT and os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "orc_kernels.hpp"))):
            if _VARIANT:
                subprocess.check_call(["make", "-C", _HERE, "liborc_%s.so" % _VARIANT], stdout=subprocess.DEVNULL)
            else:
                build()
        L = C.CDLL(so)
        L.orc_render_frame.restype = C.c_int
        L.orc_render_frame.argtypes = [C.POINTER(T.SceneDesc), C.POINTER(T.FrameParams), C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(T.Outputs), C.POINTER(T.Outputs), C.POINTER(T.Stats)]
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_build_default.argtypes = [C.c_void_p]
        L.orc_scene_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_scene_add_sphere.argtypes = [C.c_void_p, C.POINTER(T.Sphere)]
        L.orc_scene_build_sphere_instance.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(T.Affine3x4)]
        L.orc_scene_load_mesh_instance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(T.Affine3x4)]
        L.orc_scene_rebuild_tlas.argtypes = [C.c_void_p]
        L.orc_scene_set_instance_transform.argtypes = [C.c_void_p, C.c_int, C.POINTER(T.Affine3x4)]
        L.orc_scene_get_desc.argtypes = [C.c_void_p, C.POINTER(T.SceneDesc)]
        L.orc_camera_create.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(T.Camera)]
        L.orc_camera_lookat.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float, C.c_float, C.c_float, C.POINTER(T.Camera)]
        L.orc_camera_translate.argtypes = [C.POINTER(T.Camera), C.POINTER(C.c_float)]
        L.orc_camera_bake.argtypes = [C.POINTER(T.Camera), C.c_int, C.c_int]
        L.orc_sun_dir.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_rng_kat.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.orc_rng_stream.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.orc_hash3.restype = C.c_uint32
        L.orc_hash3.argtypes = [C.c_uint32] * 3
        L.orc_pack_rgba8.argtypes = [C.c_float] * 3
        L.orc_math_eval.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hit_box.argtypes = [C.c_int] + [C.c_void_p] * 6
        L.orc_dotnet_sort_by_key.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_trace_rays.argtypes = [C.POINTER(T.SceneDesc), C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 6
        L.orc_present.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_float, C.c_float, C.c_float]
        _LIB = L
    return _LIB


def _fv(v):
    return (C.c_float * 3)(*[float(x) for x in v])


MATH_FN = {"sin": 0, "cos": 1, "tan": 2, "atan": 3, "atan2": 4, "acos": 5, "asin": 6, "rsqrt": 7, "sqrt": 8,
           "fmin": 9, "fmax": 10, "floor": 11, "round": 12, "f2i": 13, "rcp": 14, "div": 15, "log": 17, "exp": 18, "pow": 19}


def math_eval(name, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    yy = None
    if y is not None:
        yy = np.ascontiguousarray(y, dtype=np.float32)
    lib().orc_math_eval(MATH_FN[name], x.size, x.ctypes.data, yy.ctypes.data if yy is not None else None, out.ctypes.data)
    return out


def hit_box(o, d, lo, hi, tmax):
    """IntersectAABB of the oracle on arrays of rays (o, d: [n,3]) and boxes (lo, hi: [n,3]); tMin = 0.001."""
    o, d, lo, hi = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3) for a in (o, d, lo, hi)]
    tm = np.ascontiguousarray(tmax, dtype=np.float32)
    out = np.zeros(len(o), np.int32)
    lib().orc_hit_box(len(o), o.ctypes.data, d.ctypes.data, lo.ctypes.data, hi.ctypes.data, tm.ctypes.data, out.ctypes.data)
    return out


def rng_kat(px, py, frame, sample, lock, salt=0xC0FFEE):
    u = (C.c_uint32 * 4)()
    f = (C.c_float * 3)()
    lib().orc_rng_kat(px, py, frame, sample, salt, lock, u, f)
    return [int(v) for v in u], [float(v) for v in f]


def camera_create(w, h, fov):
    c = T.Camera()
    lib().orc_camera_create(w, h, fov, C.byref(c))
    return c


def camera_lookat(origin, lookat, up, vfov, aspect, focus=1.0):
    c = T.Camera()
    lib().orc_camera_lookat(_fv(origin), _fv(lookat), _fv(up), vfov, aspect, focus, C.byref(c))
    return c


def camera_translate(cam, d):
    lib().orc_camera_translate(C.byref(cam), _fv(d))


def camera_bake(cam, w, h):
    lib().orc_camera_bake(C.byref(cam), w, h)


def sun_dir(az, el):
    o = (C.c_float * 3)()
    lib().orc_sun_dir(az, el, o)
    return [float(v) for v in o]


class OrcScene:
    """Oracle-side restatement of Engine/Scene.cs (host builders)."""

    def __init__(self):
        self.h = lib().orc_scene_new()

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def build_default_scene(self):
        lib().orc_scene_build_default(self.h)

    def add_texture(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = rgba.shape[:2]
        return lib().orc_scene_add_texture(self.h, w, h, rgba.ctypes.data)

    def add_sphere(self, s):
        return lib().orc_scene_add_sphere(self.h, C.byref(s))

    def build_sphere_instance(self, ids, xform=None):
        ids = list(ids)
        arr = (C.c_int * len(ids))(*ids)
        m = xform if xform is not None else T.identity_affine()
        return lib().orc_scene_build_sphere_instance(self.h, arr, len(ids), C.byref(m))

    def load_mesh_instance(self, mesh, xform=None):
        m = xform if xform is not None else T.identity_affine()
        p = mesh.ptrs()
        return lib().orc_scene_load_mesh_instance(self.h, *p, C.byref(m))

    def rebuild_tlas(self):
        lib().orc_scene_rebuild_tlas(self.h)

    def set_instance_transform(self, inst_id, xform):
        """Moves an instance (records re-derived as at creation); call rebuild_tlas() afterwards as Commit would."""
        if lib().orc_scene_set_instance_transform(self.h, int(inst_id), C.byref(xform)) != 0:
            raise IndexError("instance id out of range")

    def desc(self):
        d = T.SceneDesc()
        lib().orc_scene_get_desc(self.h, C.byref(d))
        return d

    def arrays(self):
        return T.arrays_from_scene_desc(self.desc())


def render_frame(scene_desc, params, out_struct, prev_struct=None, row_begin=0, row_end=0, run_primary=True, nthreads=None):
    """Runs PrimaryVisibilityKernel + PathTraceKernel of the oracle.  Returns Stats.
    run_primary: True both launches, False launch 2 only (G-buffer as given), 2 launch 1 only."""
    st = T.Stats()
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 64)
    rc = lib().orc_render_frame(C.byref(scene_desc), C.byref(params), row_begin, row_end, 2 if run_primary == 2 else (1 if run_primary else 0), nthreads,
                                C.byref(out_struct), C.byref(prev_struct) if prev_struct is not None else None, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render_frame failed (%d)" % rc)
    return st


def trace_rays(scene_desc, origins, dirs, brute=False):
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    n = len(o)
    t = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32); alb = np.zeros((n, 3), np.float32)
    obj = np.zeros(n, np.int32); shade = np.zeros(n, np.int32); hit = np.zeros(n, np.int32)
    lib().orc_trace_rays(C.byref(scene_desc), n, o.ctypes.data, d.ctypes.data, 1 if brute else 0,
                         t.ctypes.data, nrm.ctypes.data, alb.ctypes.data, obj.ctypes.data, shade.ctypes.data, hit.ctypes.data)
    return dict(t=t, normal=nrm, albedo=alb, objId=obj, shade=shade, hit=hit)


def present(mode, low_color, low_objid, in_w, in_h, out_w, out_h, history=None, first_frame=True, feedback=0.075, sharpness=0.10, clamp_k=1.25):
    """Oracle of the presentation step.  mode 0 = blit / bilinear upsample, 1 = TAAU (history = (color, objId) int32 arrays, updated in place)."""
    out = np.zeros(out_w * out_h, np.int32)
    lc = np.ascontiguousarray(low_color, dtype=np.int32)
    lo = np.ascontiguousarray(low_objid, dtype=np.int32) if low_objid is not None else None
    hc = history[0] if history is not None else None
    ho = history[1] if history is not None else None
    rc = lib().orc_present(mode, lc.ctypes.data, lo.ctypes.data if lo is not None else None, in_w, in_h, out.ctypes.data, out_w, out_h,
                           hc.ctypes.data if hc is not None else None, ho.ctypes.data if ho is not None else None,
                           1 if first_frame else 0, feedback, sharpness, clamp_k)
    if rc != 0:
        raise RuntimeError("orc_present failed")
    return out
