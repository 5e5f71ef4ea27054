"""Host-side mirror of the reference's Engine API for the render path, over the C ABI.

The reference's host is C# (`Engine/RTRenderer.cs`, `Engine/Scene.cs`, `Engine/SceneManager.cs`,
`Engine/Camera.cs`); there is no .NET toolchain in this image, so this module plays the
C# host's role from Python with the same class / method names and argument meaning:

    Scene.AddSphere / BuildSphereInstance / LoadObjInstance* / RebuildTLAS / BuildDefaultScene
    SceneManager.Commit  ->  RTRenderer.commit(scene)      (Scene.UploadAll, Scene.cs:258-279)
    RTRenderer.RenderDirectToPbo(pbo, w, h, frame, dt) -> RTRenderer.render_frame(w, h, frame, dt, outputs)
    RTRenderer.SetSunParams, Camera.CreateCamera / look-at ctor / Translate

(* the array-append half; the OBJ file parser is out of scope.)

Everything computed here is plumbing: the scene builders, camera math and the two kernels
all live in libhip_raytrace.so (csrc/).  There is NO fallback: if the library or a GPU is
missing, constructing RTRenderer raises.
"""
import ctypes as C
import os
import random

import numpy as np

from . import _types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRT_LIB") or os.path.join(_HERE, "csrc", "libhip_raytrace.so")   # HRT_LIB: A/B builds of the same library
_LIB = None
_HOOKS = None
HOOKS_LIB_PATH = os.environ.get("HRT_HOOKS_LIB") or os.path.join(_HERE, "csrc", "libhip_raytrace_test.so")


class HrtError(RuntimeError):
    """A C-ABI call returned a negative hrt_status."""

    def __init__(self, code, msg):
        super().__init__("hip_raytrace error %d: %s" % (code, msg))
        self.code = code


def _load(path, hooks=False):
    if not os.path.exists(path):
        raise ImportError("%s is not built. Run __graft_entry__.build(); there is no CPU fallback for the render path." % path)
    L = C.CDLL(path)
    L.hrt_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.hrt_destroy.argtypes = [C.c_void_p]
    L.hrt_destroy.restype = None
    L.hrt_last_error.argtypes = [C.c_void_p]
    L.hrt_last_error.restype = C.c_char_p
    L.hrt_scene_upload.argtypes = [C.c_void_p, C.POINTER(T.SceneDesc)]
    L.hrt_scene_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(T.BvhUpdateStats)]
    L.hrt_scene_update_positions.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(T.BvhUpdateStats)]
    L.hrt_scene_update_spheres.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(T.BvhUpdateStats)]
    L.hrt_scene_download_array.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.hrt_scene_download_tlas.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int64)]
    L.hrt_render_frame.argtypes = [C.c_void_p, C.POINTER(T.FrameParams), C.POINTER(T.RenderOpts), C.POINTER(T.Outputs), C.POINTER(T.Stats)]
    L.hrt_synchronize.argtypes = [C.c_void_p, C.POINTER(T.Stats)]
    L.hrt_present.argtypes = [C.c_void_p, C.POINTER(T.PresentParams), C.c_void_p]
    L.hrt_device_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(T.DeviceViews)]
    L.hrt_reset_history.argtypes = [C.c_void_p]
    L.hrt_frame_times.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.hrt_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.hrt_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
    L.hrt_set_workspace_limit.argtypes = [C.c_void_p, C.c_int64]
    L.hrt_device_count.restype = C.c_int
    L.hrt_version.restype = C.c_char_p
    L.hrth_scene_new.restype = C.c_void_p
    L.hrth_scene_free.argtypes = [C.c_void_p]
    L.hrth_scene_free.restype = None
    L.hrth_scene_clear.argtypes = [C.c_void_p]
    L.hrth_scene_build_default.argtypes = [C.c_void_p]
    L.hrth_scene_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.hrth_scene_add_sphere.argtypes = [C.c_void_p, C.POINTER(T.Sphere)]
    L.hrth_scene_build_sphere_instance.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(T.Affine3x4)]
    L.hrth_scene_load_mesh_instance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(T.Affine3x4)]
    L.hrth_mesh_load_obj.argtypes = [C.c_char_p, C.c_float, C.c_int, C.POINTER(C.c_void_p)]
    L.hrth_mesh_get.argtypes = [C.c_void_p, C.POINTER(T.MeshDesc)]
    L.hrth_mesh_free.argtypes = [C.c_void_p]
    L.hrth_mesh_free.restype = None
    L.hrth_mesh_material_name.argtypes = [C.c_void_p, C.c_int]
    L.hrth_mesh_material_name.restype = C.c_char_p
    L.hrth_mesh_texture_path.argtypes = [C.c_void_p, C.c_int]
    L.hrth_mesh_texture_path.restype = C.c_char_p
    L.hrth_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8))]
    L.hrth_image_free.argtypes = [C.POINTER(C.c_uint8)]
    L.hrth_image_free.restype = None
    L.hrth_scene_load_obj_instance.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(T.Affine3x4), C.c_float]
    L.hrth_last_error.restype = C.c_char_p
    L.hrth_scene_rebuild_tlas.argtypes = [C.c_void_p]
    L.hrth_scene_set_instance_transform.argtypes = [C.c_void_p, C.c_int, C.POINTER(T.Affine3x4)]
    L.hrth_scene_get_desc.argtypes = [C.c_void_p, C.POINTER(T.SceneDesc)]
    L.hrth_camera_create.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(T.Camera)]
    L.hrth_camera_lookat.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float, C.c_float, C.c_float, C.POINTER(T.Camera)]
    L.hrth_camera_translate.argtypes = [C.POINTER(T.Camera), C.POINTER(C.c_float)]
    L.hrth_camera_bake.argtypes = [C.POINTER(T.Camera), C.c_int, C.c_int]
    L.hrth_sun_dir.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
    if hooks:       # include/hrt_test_hooks.h: only in libhip_raytrace_test.so
        L.hrt_math_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hrt_math_exhaustive.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.hrt_debug_set_treelet_limits.argtypes = [C.c_int, C.c_int, C.c_int]
        L.hrt_debug_treelet_count.argtypes = [C.c_void_p]
        L.hrt_debug_treelets.argtypes = [C.POINTER(T.SceneDesc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int64)]
    return L


def lib():
    """Loads libhip_raytrace.so, the shipped library (built by `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _LIB
    if _LIB is None:
        _LIB = _load(LIB_PATH)
    return _LIB


def hooks():
    """Loads libhip_raytrace_test.so: the same sources compiled with -DHRT_TEST_HOOKS (include/hrt_test_hooks.h).  Test suite only;
    a renderer that needs a hook is created with RTRenderer(..., library=hooks())."""
    global _HOOKS
    if _HOOKS is None:
        _HOOKS = _load(HOOKS_LIB_PATH, hooks=True)
    return _HOOKS


def _fv(v):
    return (C.c_float * 3)(*[float(x) for x in v])


# ------------------------------------------------------------------ Camera (Engine/Camera.cs)
def create_camera(width, height, fov_degrees):
    """Camera.CreateCamera (Camera.cs:19-47)."""
    c = T.Camera()
    lib().hrth_camera_create(width, height, fov_degrees, C.byref(c))
    return c


def camera_look_at(origin, look_at, up, vfov_degrees, aspect, focus_dist=1.0):
    """new Camera(origin, lookAt, up, vfovDegrees, aspect, focusDist) (Camera.cs:100-119)."""
    c = T.Camera()
    lib().hrth_camera_lookat(_fv(origin), _fv(look_at), _fv(up), vfov_degrees, aspect, focus_dist, C.byref(c))
    return c


def camera_translate(cam, delta):
    """Camera.Translate (Camera.cs:121-126)."""
    lib().hrth_camera_translate(C.byref(cam), _fv(delta))


def bake_camera_derived(cam, pixel_w, pixel_h):
    """RTRenderer.BakeCameraDerived (RTRenderer.cs:241-263)."""
    lib().hrth_camera_bake(C.byref(cam), pixel_w, pixel_h)


def sun_direction(azimuth, elevation):
    o = (C.c_float * 3)()
    lib().hrth_sun_dir(azimuth, elevation, o)
    return [float(v) for v in o]


def copy_camera(cam):
    c = T.Camera()
    C.memmove(C.byref(c), C.byref(cam), C.sizeof(T.Camera))
    return c


# ------------------------------------------------------------------ mesh container (MeshHost, MeshLoaderOBJ.cs:21-31)
class MeshData:
    """Arrays of one mesh as MeshLoaderOBJ.Load would return them (MeshHost)."""

    def __init__(self, positions, triangles, texcoords, tri_uvs, materials, tri_material_index=None, textures_bgra=()):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        self.texcoords = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
        self.tri_uvs = np.ascontiguousarray(tri_uvs, dtype=np.int32).reshape(-1, 3)
        self.materials = (T.MaterialRecord * len(materials))(*materials)
        self.n_materials = len(materials)
        self.tri_mat = None if tri_material_index is None else np.ascontiguousarray(tri_material_index, dtype=np.int32)
        self.tex_w = np.array([t.shape[1] for t in textures_bgra], dtype=np.int32)
        self.tex_h = np.array([t.shape[0] for t in textures_bgra], dtype=np.int32)
        self.tex_bytes = np.concatenate([np.ascontiguousarray(t, dtype=np.uint8).reshape(-1) for t in textures_bgra]) \
            if len(textures_bgra) else np.zeros(0, np.uint8)
        self.n_tex = len(textures_bgra)
        assert len(self.tri_uvs) == len(self.triangles)

    def ptrs(self):
        return (self.positions.ctypes.data, len(self.positions), self.triangles.ctypes.data, len(self.triangles),
                self.texcoords.ctypes.data, len(self.texcoords), self.tri_uvs.ctypes.data,
                self.tri_mat.ctypes.data if self.tri_mat is not None else None, len(self.tri_mat) if self.tri_mat is not None else 0,
                C.cast(self.materials, C.c_void_p), self.n_materials,
                self.tex_w.ctypes.data if self.n_tex else None, self.tex_h.ctypes.data if self.n_tex else None,
                self.tex_bytes.ctypes.data if self.n_tex else None, self.n_tex)


class AssetFormatError(ValueError):
    """FormatException / InvalidDataException / EndOfStreamException raised by the reference's loader."""


def _raise_host(rc, what):
    msg = (lib().hrth_last_error() or b"").decode("utf-8", "replace")
    if rc == T.HRTH_ERR_NOT_FOUND:
        raise FileNotFoundError(msg or what)
    if rc == T.HRTH_ERR_FORMAT:
        raise AssetFormatError(msg or what)
    raise ValueError(msg or what)


def load_obj(path, scale=1.0, flip_winding=True):
    """MeshLoaderOBJ.Load (MeshLoaderOBJ.cs:67): OBJ + MTL + textures -> MeshData (material texture indices are local)."""
    h = C.c_void_p()
    rc = lib().hrth_mesh_load_obj(os.fsencode(path), scale, 1 if flip_winding else 0, C.byref(h))
    if rc != 0:
        _raise_host(rc, "load_obj failed")
    try:
        d = T.MeshDesc()
        lib().hrth_mesh_get(h, C.byref(d))

        def arr(ptr, n, dtype, width):
            if n == 0:
                return np.zeros((0, width), dtype)
            return np.frombuffer(C.string_at(ptr, n * width * 4), dtype).reshape(n, width).copy()

        mats = [T.MaterialRecord.from_buffer_copy(C.string_at(C.addressof(d.materials[i]), C.sizeof(T.MaterialRecord))) for i in range(d.n_materials)]
        texs, off = [], 0
        for i in range(d.n_textures):
            w, hgt = d.tex_w[i], d.tex_h[i]
            n = w * hgt * 4
            texs.append(np.frombuffer(C.string_at(C.addressof(d.tex_bgra.contents) + off, n), np.uint8).reshape(hgt, w, 4).copy()
                        if n else np.zeros((hgt, w, 4), np.uint8))
            off += n
        mesh = MeshData(arr(d.positions, d.n_positions, np.float32, 3), arr(d.triangles, d.n_triangles, np.int32, 3),
                        arr(d.texcoords, d.n_texcoords, np.float32, 2), arr(d.tri_uvs, d.n_triangles, np.int32, 3), mats,
                        arr(d.tri_material_index, d.n_tri_material_index, np.int32, 1).reshape(-1), texs)
        mesh.material_names = [(lib().hrth_mesh_material_name(h, i) or b"").decode("utf-8", "replace") for i in range(d.n_materials)]
        mesh.texture_paths = [(lib().hrth_mesh_texture_path(h, i) or b"").decode("utf-8", "replace") for i in range(d.n_textures)]
        return mesh
    finally:
        lib().hrth_mesh_free(h)


def load_image(path):
    """LoadTextureBGRA (MeshLoaderOBJ.cs:456-593): .tga (raw / RLE, 8/24/32 bit), .png (<= 8 bits per sample) or uncompressed .bmp -> (H, W, 4) uint8 BGRA, row 0 = top."""
    w, h, p = C.c_int(), C.c_int(), C.POINTER(C.c_uint8)()
    rc = lib().hrth_image_load(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p))
    if rc != 0:
        _raise_host(rc, "load_image failed")
    try:
        n = w.value * h.value * 4
        return np.frombuffer(C.string_at(p, n), np.uint8).reshape(h.value, w.value, 4).copy()
    finally:
        lib().hrth_image_free(p)


# ------------------------------------------------------------------ Scene (Engine/Scene.cs, host lists + builders)
class Scene:
    def __init__(self):
        self._h = lib().hrth_scene_new()

    def __del__(self):
        try:
            if self._h:
                lib().hrth_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def build_default_scene(self):
        """Scene.BuildDefaultScene (Scene.cs:83-142)."""
        lib().hrth_scene_build_default(self._h)

    def add_texture(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = rgba.shape[:2]
        r = lib().hrth_scene_add_texture(self._h, w, h, rgba.ctypes.data)
        if r < 0:
            raise ValueError("add_texture: invalid texture")
        return r

    def add_sphere(self, sphere):
        """Scene.AddSphere (Scene.cs:315-321)."""
        return lib().hrth_scene_add_sphere(self._h, C.byref(sphere))

    def build_sphere_instance(self, sphere_ids, object_to_world=None):
        """Scene.BuildSphereInstance (Scene.cs:323-356); the record is appended to the instance list."""
        ids = list(sphere_ids)
        arr = (C.c_int * len(ids))(*ids)
        m = object_to_world if object_to_world is not None else T.identity_affine()
        r = lib().hrth_scene_build_sphere_instance(self._h, arr, len(ids), C.byref(m))
        if r < 0:
            raise ValueError("build_sphere_instance: invalid sphere ids")
        return r

    def load_mesh_instance(self, mesh, object_to_world=None):
        """Scene.LoadObjInstance after MeshLoaderOBJ.Load (Scene.cs:151-256)."""
        m = object_to_world if object_to_world is not None else T.identity_affine()
        r = lib().hrth_scene_load_mesh_instance(self._h, *mesh.ptrs(), C.byref(m))
        if r < 0:
            raise ValueError("load_mesh_instance: invalid mesh arrays")
        return r

    def load_obj_instance(self, obj_path, object_to_world=None, uniform_scale=1.0):
        """Scene.LoadObjInstance(objPath, objectToWorld, uniformScale) (Scene.cs:144-256)."""
        m = object_to_world if object_to_world is not None else T.identity_affine()
        r = lib().hrth_scene_load_obj_instance(self._h, os.fsencode(obj_path), C.byref(m), uniform_scale)
        if r < 0:
            _raise_host(r, "load_obj_instance failed")
        return r

    def rebuild_tlas(self):
        """Scene.RebuildTLAS (Scene.cs:358-368)."""
        lib().hrth_scene_rebuild_tlas(self._h)

    def set_instance_transform(self, inst_id, xform):
        """Moves an instance of the host scene (records re-derived as at creation); follow with rebuild_tlas() + commit,
        or mirror the move on the device with RTRenderer.update_instances."""
        if lib().hrth_scene_set_instance_transform(self._h, int(inst_id), C.byref(xform)) != 0:
            raise IndexError("instance id out of range")

    def desc(self):
        d = T.SceneDesc()
        lib().hrth_scene_get_desc(self._h, C.byref(d))
        return d

    def arrays(self):
        return T.arrays_from_scene_desc(self.desc())


# ------------------------------------------------------------------ RTRenderer (Engine/RTRenderer.cs)
class RTRenderer:
    """Frame orchestration of RTRenderer.RenderDirectToPbo up to the end-of-frame Synchronize
    (RTRenderer.cs:105-205,233).  Presentation (PBO map, TAAU/blit) is out of scope."""

    def __init__(self, device_ids=None, width=1280, height=720, build_default_scene=False, library=None):
        L = self._L = library or lib()
        ids = list(device_ids) if device_ids is not None else [0]
        arr = (C.c_int * len(ids))(*ids)
        h = C.c_void_p()
        rc = L.hrt_create(arr, len(ids), C.byref(h))
        if rc != 0:
            raise HrtError(rc, (L.hrt_last_error(None) or b"").decode())
        self._ctx = h
        self.n_devices = len(ids)
        # private fields of the reference's RTRenderer (RTRenderer.cs:43-61), same defaults except
        # render scale (benchmarks render at 1.0) -- callers set what they need
        self.enable_temporal_reuse = 1
        self.enable_spatial_reuse = 1
        self.rng_lock_noise = 1
        self.spp = 2
        self.max_depth = 3
        self.sun_azimuth = 0.0
        self.sun_elevation = 0.9
        self.sun_speed = 0.0
        self.dir_light_radiance = (10.0, 10.0, 10.0)
        self.sky_tint_top = (0.5, 0.7, 1.0)
        self.sky_tint_bottom = (1.0, 1.0, 1.0)
        self.camera = create_camera(max(1, width), max(1, height), 60.0)
        camera_translate(self.camera, (1.0, 0.0, -4.0))           # RTRenderer.cs:78-79
        self.prev_camera = copy_camera(self.camera)
        self.scene = None
        self.last_params = None
        if build_default_scene:
            s = Scene()
            s.build_default_scene()
            self.commit(s)

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.hrt_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise HrtError(rc, (self._L.hrt_last_error(self._ctx) or b"").decode())

    def commit(self, scene_or_desc):
        """SceneManager.Commit -> BvhManager.BuildOrRefit -> Scene.UploadAll."""
        if isinstance(scene_or_desc, T.SceneDesc):
            d = scene_or_desc
        else:
            self.scene = scene_or_desc
            d = scene_or_desc.desc()
        self._check(self._L.hrt_scene_upload(self._ctx, C.byref(d)))

    def update_instances(self, ids, transforms, policy=T.REBUILD_AUTO):
        """BvhManager.BuildOrRefit(scene, policy) for moved instances, on the device (hrt_scene_update_instances).
        ids: instance ids; transforms: Affine3x4 objects or an (n, 12) float32 array, row-major 3x4.  Returns BvhUpdateStats."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if len(transforms) and isinstance(transforms[0], T.Affine3x4):
            xf = np.array([[getattr(a, f"m{r}{c}") for r in range(3) for c in range(4)] for a in transforms], dtype=np.float32)
        else:
            xf = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 12)
        if xf.shape[0] != ids.size:
            raise ValueError("one transform per instance id")
        st = T.BvhUpdateStats()
        self._check(self._L.hrt_scene_update_instances(self._ctx, ids.ctypes.data if ids.size else None, ids.size,
                                                     xf.ctypes.data if ids.size else None, policy, C.byref(st)))
        return st

    def update_positions(self, first_vertex, positions, policy=T.REBUILD_AUTO):
        """Deforming meshes: meshPositions[first_vertex : first_vertex + n] := positions ((n, 3) float32); every triangle-mesh
        BLAS is refitted on the device, then the TLAS per `policy` (hrt_scene_update_positions).  Returns BvhUpdateStats."""
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        st = T.BvhUpdateStats()
        self._check(self._L.hrt_scene_update_positions(self._ctx, int(first_vertex), pos.shape[0], pos.ctypes.data if pos.size else None,
                                                     policy, C.byref(st)))
        return st

    def update_spheres(self, first_sphere, spheres, policy=T.REBUILD_AUTO):
        """spheres[first_sphere : first_sphere + n] := spheres (a list of T.Sphere or a structured numpy array); sphere-set BLASes
        are refitted on the device, then the TLAS per `policy` (hrt_scene_update_spheres).  Returns BvhUpdateStats."""
        if isinstance(spheres, np.ndarray):
            buf = np.ascontiguousarray(spheres)
            n, ptr = len(buf), buf.ctypes.data
        else:
            buf = (T.Sphere * max(1, len(spheres)))(*spheres)
            n, ptr = len(spheres), C.addressof(buf)
        st = T.BvhUpdateStats()
        self._check(self._L.hrt_scene_update_spheres(self._ctx, int(first_sphere), n, ptr if n else None, policy, C.byref(st)))
        return st

    def download_array(self, name, slot=0):
        """One of the 15 scene arrays as it is on the device now (numpy structured array / int32)."""
        names = [n for n, _ in T.SCENE_ARRAYS]
        k = names.index(name)
        cnt = C.c_int64()
        self._check(self._L.hrt_scene_download_array(self._ctx, slot, k, None, 0, C.byref(cnt)))
        out = np.zeros(max(1, cnt.value), dtype=T.np_dtype(T.SCENE_ARRAYS[k][1]))
        self._check(self._L.hrt_scene_download_array(self._ctx, slot, k, out.ctypes.data, len(out), C.byref(cnt)))
        return out[:cnt.value]

    def download_tlas(self, slot=0):
        """(tlasNodes, tlasInstanceIndices, instances) of the TLAS in use, as ctypes arrays in the reference's layout."""
        cnt = (C.c_int64 * 3)()
        self._check(self._L.hrt_scene_download_tlas(self._ctx, slot, None, 0, None, 0, None, 0, cnt))
        nodes = (T.BvhNode * max(1, cnt[0]))()
        idx = (C.c_int32 * max(1, cnt[1]))()
        inst = (T.InstanceRecord * max(1, cnt[2]))()
        self._check(self._L.hrt_scene_download_tlas(self._ctx, slot, nodes, cnt[0], idx, cnt[1], inst, cnt[2], cnt))
        return nodes, idx, inst, tuple(cnt)

    def set_sun_params(self, speed_rad_per_sec, elevation_rad):
        """RTRenderer.SetSunParams (RTRenderer.cs:99-103)."""
        self.sun_speed = speed_rad_per_sec
        self.sun_elevation = elevation_rad

    def make_params(self, width, height, frame, dt=0.0):
        """Parameter assembly of RenderDirectToPbo (RTRenderer.cs:109-202) at render scale 1."""
        w, h = max(1, width), max(1, height)
        bake_camera_derived(self.camera, w, h)
        bake_camera_derived(self.prev_camera, w, h)
        temporal_seed = 0 if self.rng_lock_noise == 0 else random.randint(-2 ** 31, 2 ** 31 - 2)
        dtc = min(max(dt, 0.0), 0.1)
        self.sun_azimuth += self.sun_speed * dtc
        two_pi = 6.28318530717958647692
        if self.sun_azimuth >= two_pi:
            self.sun_azimuth -= two_pi
        elif self.sun_azimuth < 0.0:
            self.sun_azimuth += two_pi
        p = T.FrameParams()
        p.width, p.height, p.frame = w, h, frame
        p.cam = copy_camera(self.camera)
        p.prevCam = copy_camera(self.prev_camera)
        p.dirLightDir = T.f3(*sun_direction(self.sun_azimuth, self.sun_elevation))
        p.dirLightRadiance = T.f3(*self.dir_light_radiance)
        p.skyTintTop = T.f3(*self.sky_tint_top)
        p.skyTintBottom = T.f3(*self.sky_tint_bottom)
        p.debugCamSeq = 0
        p.enableTemporalReuse = self.enable_temporal_reuse
        p.enableSpatialReuse = self.enable_spatial_reuse
        p.rngLockNoise = temporal_seed
        p.spp = self.spp
        p.maxDepth = self.max_depth
        return p

    def render_params(self, params, outputs=None, flags=0, rows=None, strips=None):
        """The two launches + sync for an explicit FrameParams.  Returns Stats.
        rows=(y0,y1) restricts the frame to a row range, strips=(n,i) to every n-th 8-row strip of it;
        flags & FLAG_NO_SYNC only enqueues (collect with synchronize())."""
        st = T.Stats()
        opts = T.RenderOpts(flags, rows[0] if rows else 0, rows[1] if rows else 0,
                            strips[0] if strips else 1, strips[1] if strips else 0)
        self._check(self._L.hrt_render_frame(self._ctx, C.byref(params), C.byref(opts),
                                           C.byref(outputs) if outputs is not None else None, C.byref(st)))
        self.last_params = params
        return st

    def render_frame(self, width, height, frame, dt=0.0, outputs=None, flags=0, rows=None):
        """RenderDirectToPbo(pbo, width, height, frame, dt) without the presentation step."""
        p = self.make_params(width, height, frame, dt)
        st = self.render_params(p, outputs, flags, rows)
        self.prev_camera = copy_camera(self.camera)             # RTRenderer.cs:236
        return st

    def synchronize(self):
        """Waits for frames enqueued with FLAG_NO_SYNC; Stats.kernel_ms are sums over Stats.frames frames."""
        st = T.Stats()
        self._check(self._L.hrt_synchronize(self._ctx, C.byref(st)))
        return st

    def frame_times(self, launch=1, slot=0):
        """Per-frame HIP-event times (ms) of the frames the last synchronize() / blocking frame collected."""
        n = C.c_int(0)
        self._check(self._L.hrt_frame_times(self._ctx, slot, launch, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), np.float32)
        self._check(self._L.hrt_frame_times(self._ctx, slot, launch, out.ctypes.data, n.value, None))
        return out[:n.value]

    def register_host(self, arrays):
        """Page-locks host arrays used as gather targets (hrt_host_register); arrays: dict or iterable of numpy arrays."""
        for a in (arrays.values() if isinstance(arrays, dict) else arrays):
            self._check(self._L.hrt_host_register(self._ctx, a.ctypes.data, a.nbytes))

    def unregister_host(self, arrays):
        for a in (arrays.values() if isinstance(arrays, dict) else arrays):
            self._check(self._L.hrt_host_unregister(self._ctx, a.ctypes.data))

    def set_workspace_limit(self, max_resident_paths):
        """Caps the streamed pipeline's path workspace (0 = default): larger frames run in sample batches, same results."""
        self._check(self._L.hrt_set_workspace_limit(self._ctx, int(max_resident_paths)))

    def present(self, out_width, out_height, taau=True, out=None, feedback=0.0, sharpness=0.0, clamp_k=0.0):
        """Presentation step of RenderDirectToPbo (RTRenderer.cs:208-231): TAAU resolve, or blit / bilinear upsample.
        Returns the display-size packed colour as an int32 array."""
        pp = T.PresentParams(out_width, out_height, T.PRESENT_TAAU if taau else T.PRESENT_RESAMPLE, feedback, sharpness, clamp_k)
        if out is None:
            out = np.zeros(out_width * out_height, np.int32)
        self._check(self._L.hrt_present(self._ctx, C.byref(pp), out.ctypes.data))
        return out

    def render_direct(self, out_width, out_height, frame, dt=0.0, render_scale=0.67, taau=True, flags=0):
        """RenderDirectToPbo(pbo, width, height, frame, dt) end to end: internal size = round(out * renderScale)
        (RTRenderer.cs:113-116), the two launches, then the presentation step.  Returns (display colour, Stats)."""
        in_w = max(1, int(np.rint(np.float32(out_width) * np.float32(render_scale))))
        in_h = max(1, int(np.rint(np.float32(out_height) * np.float32(render_scale))))
        st = self.render_frame(in_w, in_h, frame, dt, None, flags)
        return self.present(out_width, out_height, taau), st

    def reset_history(self):
        self._check(self._L.hrt_reset_history(self._ctx))

    def device_views(self, slot=0):
        v = T.DeviceViews()
        self._check(self._L.hrt_device_buffers(self._ctx, slot, C.byref(v)))
        return v

    def math_exhaustive(self, which):
        """(mismatches, bits of the first one) of a trimmed device function against its IEEE definition over its whole domain."""
        n, first = C.c_uint64(0), C.c_uint32(0)
        self._check(self._L.hrt_math_exhaustive(self._ctx, which, C.byref(n), C.byref(first)))
        return n.value, first.value

    def math_probe(self, fn, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        yy = np.ascontiguousarray(y, dtype=np.float32) if y is not None else None
        self._check(self._L.hrt_math_probe(self._ctx, fn, x.size, x.ctypes.data, yy.ctypes.data if yy is not None else None, out.ctypes.data))
        return out


# ------------------------------------------------------------------ SceneManager / BvhManager (Engine/SceneManager.cs, BvhManager.cs)
class SceneManager:
    """SceneManager (SceneManager.cs:12-38) over one RTRenderer: Scene, BuildDefaultScene, LoadObjInstance, Commit(policy),
    ReplaceScene.  Commit is BvhManager.BuildOrRefit (BvhManager.cs:27) with the RebuildPolicy honoured: the first commit, and
    any commit after the scene's structure changed (instances, spheres, meshes added), uploads everything as the reference
    does; a commit after only instance transforms moved (set_instance_transform) updates the device copy in place."""

    def __init__(self, renderer, existing_scene=None):
        if renderer is None:
            raise ValueError("renderer is None")                    # ArgumentNullException(nameof(cuda))
        self._r = renderer
        self._scene = existing_scene if existing_scene is not None else Scene()
        self._uploaded_shape = None
        self._moved = {}
        self.last_update = None

    @property
    def scene(self):
        return self._scene

    def build_default_scene(self):
        self._scene.build_default_scene()

    def load_obj_instance(self, obj_path, object_to_world=None, uniform_scale=1.0):
        return self._scene.load_obj_instance(obj_path, object_to_world, uniform_scale)

    def set_instance_transform(self, inst_id, object_to_world):
        """Moves an instance of the scene; takes effect at the next commit."""
        self._scene.set_instance_transform(inst_id, object_to_world)
        self._moved[int(inst_id)] = object_to_world

    def _shape(self):
        d = self._scene.desc()
        return tuple(getattr(d, "n_" + n) for n, _ in T.SCENE_ARRAYS)

    def commit(self, policy=T.REBUILD_AUTO):
        shape = self._shape()
        if self._uploaded_shape != shape:
            self._scene.rebuild_tlas()                                # a host that added or moved things before its first commit
            self._r.commit(self._scene)
            self._uploaded_shape = self._shape()
            self.last_update = None
        elif self._moved:
            ids = sorted(self._moved)
            self.last_update = self._r.update_instances(ids, [self._moved[i] for i in ids], policy)
        self._moved = {}

    def replace_scene(self, new_scene, rebuild_immediately=True, policy=T.REBUILD_AUTO):
        if new_scene is None:
            raise ValueError("new_scene is None")
        self._scene = new_scene
        self._uploaded_shape = None
        self._moved = {}
        if rebuild_immediately:
            self.commit(policy)


def device_count():
    return lib().hrt_device_count()
