"""oracle/orc_indep.py -- a SECOND, independent restatement of the reference's two kernels.  TEST INFRASTRUCTURE ONLY.

oracle/orc_kernels.hpp (C++) is what the HIP path is compared with; the reference has no fixtures and cannot be run here, so
nothing reference-side pins that restatement (parity unpinned).  This file reads the reference's device code a second
time, in another language and another session, straight from the C# (file:line below, paths under
/root/reference/ILGPU_Raytracing/Engine/), as scalar numpy.float32 arithmetic in the reference's own statement order -- not from
orc_kernels.hpp.  tests/test_oracle_indep.py requires both restatements to agree bit for bit on every output array of small frames:
a transcription slip in either one shows up as a mismatch (a shared misreading of the C# would not, which is why parity stays
"unpinned" in the strict sense).

Scope: everything the two kernels execute -- sphere sets and triangle meshes, instance transforms, textures (sphere and triangle
diffuse maps, alpha cut-outs with the point / linear band of the any-hit walk), mirror / glass / Lambert vertices, ReSTIR-DI with
temporal and spatial reuse, Russian roulette -- at any spp / maxDepth.  Transcendentals (XMath.Sin / Cos / Tan / Atan2 / Acos) come
from the shared arithmetic contract include/hrt_math.h through a callback: ILGPU.Algorithms' own bits are unknowable here, so that
boundary is shared by definition.
"""
import numpy as np

f32 = np.float32
U32 = 0xFFFFFFFF
U64 = 0xFFFFFFFFFFFFFFFF
PI = f32(3.14159265358979323846)          # RTRay.cs:183
INV_PI = f32(0.31830988618379067154)      # RTRay.cs:184
EPS_N = f32(0.0025)                       # RTRay.cs:185
EPS_MIN = f32(1e-6)                       # RTRay.cs:186
SHADING_MIRROR, SHADING_GLASS = 1, 2      # Sphere.cs


# ---------------------------------------------------------------- Float3 (Float3.cs:6-114), as tuples of float32
def v3(x, y, z): return (f32(x), f32(y), f32(z))
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def muls(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def mulv(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
def neg(a): return (-a[0], -a[1], -a[2])
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]                                            # :88-91
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])  # :82-85


def fmin(a, b):      # XMath.Min on floats: the smaller; -0 before +0
    if a < b: return a
    if b < a: return b
    if a == b: return a if np.signbit(a) else b
    return b if a != a else a


def fmax(a, b):
    if a > b: return a
    if b > a: return b
    if a == b: return b if np.signbit(a) else a
    return b if a != a else a


def rsqrt(x): return f32(1.0) / np.sqrt(x)                      # XMath.Rsqrt on a CPU device: 1 / Sqrt


def normalize(v):                                               # Float3.cs:94-98
    inv = rsqrt(fmax(f32(1e-20), v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
    return (v[0] * inv, v[1] * inv, v[2] * inv)


def f3(rec): return (f32(rec["X"]), f32(rec["Y"]), f32(rec["Z"]))


def trunc_int(x):                                               # C# (int)float: toward zero; NaN and out-of-range as include/hrt_math.h hrt_f2i
    if not (x >= -2147483648.0 and x < 2147483648.0):
        return -2147483648
    return int(x)


# ---------------------------------------------------------------- RNG (RTUtils.cs:20-138)
class RNG:
    def __init__(self, seed):                                   # Create :25-30
        self.state = 1 if seed == 0 else seed & U32

    def next_uint(self):                                        # :33-42
        x = self.state
        x ^= (x << 13) & U32
        x ^= x >> 17
        x ^= (x << 5) & U32
        self.state = x if x != 0 else 1
        return self.state

    def next_float(self):                                       # :45-49
        u = self.next_uint()
        return f32(u & 0x00FFFFFF) * f32(1.0 / 16777216.0)


def _splitmix32(x):                                             # :54-62
    x = (x + 0x9E3779B97F4A7C15) & U64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & U64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & U64
    x ^= x >> 31
    return (x ^ (x >> 32)) & U32


def _pcg_permute(x):                                            # :65-74
    x ^= x >> 16
    x = (x * 0x7FEB352D) & U32
    x ^= x >> 15
    x = (x * 0x846CA68B) & U32
    x ^= x >> 16
    return x


def _hash32(x):                                                 # :77-84 (== RTRay.Hash :637-641)
    x ^= x >> 17; x = (x * 0xED5AD4BB) & U32
    x ^= x >> 11; x = (x * 0xAC4C1B51) & U32
    x ^= x >> 15; x = (x * 0x31848BAB) & U32
    x ^= x >> 14
    return x


def _rotl(v, r): return ((v << (r & 31)) | (v >> ((32 - r) & 31))) & U32          # :100-103


def _make_seed32(a, b, c, d):                                   # :87-97
    lane0 = ((a << 32) | b) & U64
    lane1 = ((c << 32) | d) & U64
    s0 = _splitmix32(lane0 ^ 0xD1B54A32D192ED03)
    s1 = _splitmix32(lane1 ^ 0x94D049BB133111EB)
    s = _pcg_permute(s0 ^ ((_rotl(s1, 13) + 0x9E3779B1) & U32))
    return s | 1


def rng_from_index(index, width, height, frame, sample, salt, lock_noise):        # CreateFromIndex1D :108-113 + CreateFromPixel :116-137
    px = (index % max(1, width)) & U32
    py = (index // max(1, width)) & U32
    f = 0 if lock_noise != 0 else frame & U32
    ln = lock_noise & U32
    ln_mix0 = (_hash32(ln) ^ ((ln * 0x1B873593) & U32)) if lock_noise != 0 else 0
    ln_mix1 = ((_rotl(ln, 7) * 0x85EBCA6B) & U32) if lock_noise != 0 else 0
    lane0a = px ^ 0xB5297A4D
    lane0b = ((py * 0x68E31DA4) & U32) ^ ((f * 0x9E3779B1 + 0x85EBCA6B) & U32) ^ ln_mix0
    lane1a = ((sample ^ 0xC2B2AE35) + _rotl(px, 16)) & U32
    lane1b = (((salt ^ 0x27D4EB2F) + _rotl(py, 8)) & U32) ^ ln_mix1          # '+' binds tighter than '^' (:131)
    return RNG(_make_seed32(lane0a, lane0b, lane1a, lane1b))


# ---------------------------------------------------------------- rays (RTUtils.cs:6-18, RTRay.cs:548-558)
def inv_dir(d):
    one = f32(1.0)
    return (one / (d[0] if d[0] != 0 else f32(1e-8)), one / (d[1] if d[1] != 0 else f32(1e-8)), one / (d[2] if d[2] != 0 else f32(1e-8)))


def generate_ray(cam, u, v):                                    # Ray.GenerateRay RTUtils.cs:13-17
    d = normalize(sub(add(add(cam["lowerLeft"], muls(cam["horizontal"], u)), muls(cam["vertical"], v)), cam["origin"]))
    return (cam["origin"], d, inv_dir(d))


def make_ray_with_normal_offset(origin, n, direction):          # RTRay.cs:552-558
    d = normalize(direction)
    s = f32(1.0) if dot(n, d) >= 0 else f32(-1.0)
    o = add(origin, muls(n, EPS_N * s))
    return (o, d, inv_dir(d))


# ---------------------------------------------------------------- intersection (SceneDeviceViews.cs:496-537)
def intersect_aabb(ray, bmin, bmax, t_min, t_max):
    o, _, inv = ray
    t1 = (bmin[0] - o[0]) * inv[0]
    t2 = (bmax[0] - o[0]) * inv[0]
    tmin = fmin(t1, t2)
    tmax = fmax(t1, t2)
    t1 = (bmin[1] - o[1]) * inv[1]
    t2 = (bmax[1] - o[1]) * inv[1]
    tmin = fmax(tmin, fmin(t1, t2))
    tmax = fmin(tmax, fmax(t1, t2))
    t1 = (bmin[2] - o[2]) * inv[2]
    t2 = (bmax[2] - o[2]) * inv[2]
    tmin = fmax(tmin, fmin(t1, t2))
    tmax = fmin(tmax, fmax(t1, t2))
    return bool(tmax >= fmax(tmin, t_min)) and bool(tmin <= t_max)


def intersect_sphere(ray, center, radius):
    """-> (hit, t, n)"""
    o, d, _ = ray
    oc = sub(o, center)
    a = dot(d, d)
    b = f32(2.0) * dot(oc, d)
    c = dot(oc, oc) - radius * radius
    disc = b * b - f32(4.0) * a * c
    if disc < 0:
        return False, f32(0), None
    sqrt_d = np.sqrt(disc)
    t0 = (-b - sqrt_d) / (f32(2.0) * a)
    t1 = (-b + sqrt_d) / (f32(2.0) * a)
    t = t0
    if t < f32(0.001):
        t = t1
        if t < f32(0.001):
            return False, t, None
    p = add(o, muls(d, t))
    return True, t, normalize(sub(p, center))


def intersect_triangle(ray, v0, v1, v2):                       # IntersectTriangleMT_Bary :540-558 -> (hit, t, n, bu, bv)
    o, d, _ = ray
    e1 = sub(v1, v0)
    e2 = sub(v2, v0)
    p = cross(d, e2)
    det = dot(e1, p)
    if abs(det) < f32(1e-8):
        return False, f32(0), None, f32(0), f32(0)
    inv_det = f32(1.0) / det
    tv = sub(o, v0)
    bu = dot(tv, p) * inv_det
    if bu < 0 or bu > 1:
        return False, f32(0), None, bu, f32(0)
    q = cross(tv, e1)
    bv = dot(d, q) * inv_det
    if bv < 0 or bu + bv > 1:
        return False, f32(0), None, bu, bv
    t = dot(e2, q) * inv_det
    if t <= 0:
        return False, t, None, bu, bv
    return True, t, normalize(cross(e1, e2)), bu, bv


def transform_point(m, p):                                      # :483-487
    return (m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3], m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7], m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11])


def transform_vector(m, v):                                     # :489-493
    return (m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[4] * v[0] + m[5] * v[1] + m[6] * v[2], m[8] * v[0] + m[9] * v[1] + m[10] * v[2])


def transform_ray(m, w):                                        # :475-481
    d = transform_vector(m, w[1])
    return (transform_point(m, w[0]), d, inv_dir(d))


class Views:
    """SceneDeviceViews for sphere-set scenes (SceneDeviceViews.cs:13-27), unpacked once from the numpy arrays."""

    def __init__(self, arrs):
        def node(n):
            return (f3(n["boundsMin"]), f3(n["boundsMax"]), int(n["left"]), int(n["first"]), int(n["count"]), int(n["skipIndex"]))
        self.tlas = [node(n) for n in arrs["tlasNodes"]]
        self.blas = [node(n) for n in arrs["blasNodes"]]
        self.tlas_inst = [int(v) for v in arrs["tlasInstanceIndices"]]
        self.sphere_prim = [int(v) for v in arrs["spherePrimIdx"]]
        aff = lambda m: tuple(f32(m[k]) for k in ("m00", "m01", "m02", "m03", "m10", "m11", "m12", "m13", "m20", "m21", "m22", "m23"))
        self.inst = [(int(r["blasRoot"]), int(r["blasNodeCount"]), aff(r["objectToWorld"]), aff(r["worldToObject"]), f32(r["uniformScale"]), int(r["type"]))
                     for r in arrs["instances"]]
        mat = lambda m: dict(Kd=f3(m["Kd"]), HasDiffuseMap=int(m["HasDiffuseMap"]), DiffuseTexIndex=int(m["DiffuseTexIndex"]), HasAlphaMap=int(m["HasAlphaMap"]),
                             AlphaTexIndex=int(m["AlphaTexIndex"]), TwoSided=int(m["TwoSided"]), AlphaCutoff=f32(m["AlphaCutoff"]))
        self.spheres = [(f3(s["center"]), f32(s["radius"]), f3(s["albedo"]), f3(s["material"]["Kd"]), int(s["shading"]), f32(s["ior"]), mat(s["material"]))
                        for s in arrs["spheres"]]
        self.math = None                                                              # set by render(): the shared transcendental functions
        self.cover = {"cutout_closest": 0, "any_point_below": 0, "any_point_above": 0, "any_band_linear": 0, "sphere_texture": 0, "two_sided_flip": 0}   # branch coverage, for the tests
        self.tri_prim = [int(v) for v in arrs["triPrimIdx"]]
        self.positions = [f3(v) for v in arrs["meshPositions"]]
        self.tris = [(int(t["i0"]), int(t["i1"]), int(t["i2"])) for t in arrs["meshTris"]]
        self.texcoords = [(f32(t["X"]), f32(t["Y"])) for t in arrs["meshTexcoords"]] or [(f32(0), f32(0))]      # AllocateOrEmpty: one zeroed element (Scene.cs:370-377)
        self.tri_uvs = [(int(t["t0"]), int(t["t1"]), int(t["t2"])) for t in arrs["meshTriUVs"]]
        self.tri_mat = [int(v) for v in arrs["triMatIndex"]]
        zero_mat = dict(Kd=v3(0, 0, 0), HasDiffuseMap=0, DiffuseTexIndex=0, HasAlphaMap=0, AlphaTexIndex=0, TwoSided=0, AlphaCutoff=f32(0))
        self.materials = [mat(m) for m in arrs["materials"]] or [zero_mat]
        self.texels = [(int(t["R"]), int(t["G"]), int(t["B"]), int(t["A"])) for t in arrs["texels"]] or [(0, 0, 0, 0)]
        self.tex_infos = [(int(t["Offset"]), int(t["Width"]), int(t["Height"])) for t in arrs["texInfos"]] or [(0, 0, 0)]

    # ---- texture sampling (SceneDeviceViews.cs:330-472)
    def texel_raw(self, info, x, y):                                                 # :330-339
        off, w, h = info
        if w <= 0 or h <= 0:
            return (0, 0, 0, 0)
        sx = max(0, min(w - 1, x))
        sy = max(0, min(h - 1, y))
        return self.texels[off + sy * w + sx]

    @staticmethod
    def _c255(v): return f32(v) * (f32(1.0) / f32(255.0))

    def luma01(self, p):                                                             # :342-348
        r, g, b = self._c255(p[0]), self._c255(p[1]), self._c255(p[2])
        return f32(0.2126) * r + f32(0.7152) * g + f32(0.0722) * b

    def _taps(self, info, u, v):                                                     # the common head of the three bilinear samplers
        _, w, h = info
        fu = u - np.floor(u)
        fv = f32(1.0) - (v - np.floor(v))
        x = fu * f32(w - 1)
        y = fv * f32(h - 1)
        x0 = trunc_int(np.floor(x)); y0 = trunc_int(np.floor(y))
        x1 = min(w - 1, x0 + 1); y1 = min(h - 1, y0 + 1)
        return x0, y0, x1, y1, x - f32(x0), y - f32(y0)

    def sample_texture_linear(self, info, u, v):                                     # :358-385 (and the RGB half of :431-472)
        if info[1] <= 0 or info[2] <= 0:
            return v3(1, 1, 1)
        x0, y0, x1, y1, tx, ty = self._taps(info, u, v)
        rgb = lambda p: (self._c255(p[0]), self._c255(p[1]), self._c255(p[2]))
        c00, c10 = rgb(self.texel_raw(info, x0, y0)), rgb(self.texel_raw(info, x1, y0))
        c01, c11 = rgb(self.texel_raw(info, x0, y1)), rgb(self.texel_raw(info, x1, y1))
        cx0 = add(muls(c00, f32(1.0) - tx), muls(c10, tx))
        cx1 = add(muls(c01, f32(1.0) - tx), muls(c11, tx))
        return add(muls(cx0, f32(1.0) - ty), muls(cx1, ty))

    def sample_mask_linear(self, info, u, v):                                        # :388-415
        if info[1] <= 0 or info[2] <= 0:
            return f32(1.0)
        x0, y0, x1, y1, tx, ty = self._taps(info, u, v)
        a00, a10 = self.luma01(self.texel_raw(info, x0, y0)), self.luma01(self.texel_raw(info, x1, y0))
        a01, a11 = self.luma01(self.texel_raw(info, x0, y1)), self.luma01(self.texel_raw(info, x1, y1))
        ax0 = a00 * (f32(1.0) - tx) + a10 * tx
        ax1 = a01 * (f32(1.0) - tx) + a11 * tx
        return ax0 * (f32(1.0) - ty) + ax1 * ty

    def sample_mask_point(self, info, u, v):                                         # :418-428 (XMath.Round: half to even)
        _, w, h = info
        if w <= 0 or h <= 0:
            return f32(1.0)
        fu = u - np.floor(u)
        fv = f32(1.0) - (v - np.floor(v))
        x = trunc_int(np.rint(fu * f32(w - 1)))
        y = trunc_int(np.rint(fv * f32(h - 1)))
        return self.luma01(self.texel_raw(info, x, y))

    def tri_uv(self, tri_index, bu, bv):
        t0, t1, t2 = (self.texcoords[i] for i in self.tri_uvs[tri_index])
        w = f32(1.0) - bu - bv
        return t0[0] * w + t1[0] * bu + t2[0] * bv, t0[1] * w + t1[1] * bu + t2[1] * bv

    # TraverseBLAS_Tri_Textured :173-237 -> (hit, tClosest, nObj, albedo, tri)
    def traverse_blas_tri(self, ray, start, end):
        t_closest, n_obj, albedo, tri_out = f32(1e30), v3(0, 0, 0), v3(0.85, 0.85, 0.85), -1
        cur = start
        while cur != -1 and cur < end:
            bmin, bmax, left, first, count, skip = self.blas[cur]
            if intersect_aabb(ray, bmin, bmax, f32(0.001), t_closest):
                if count > 0:
                    for i in range(first, first + count):
                        ti = self.tri_prim[i]
                        v0, v1, v2 = (self.positions[k] for k in self.tris[ti])
                        hit, t, nn, bu, bv = intersect_triangle(ray, v0, v1, v2)
                        if not hit:
                            continue
                        m = self.materials[self.tri_mat[ti]]
                        if t > f32(0.001) and t < t_closest:
                            uu, vv = self.tri_uv(ti, bu, bv)
                            alpha, kd = f32(1.0), m["Kd"]
                            if m["HasDiffuseMap"] != 0 and 0 <= m["DiffuseTexIndex"] < len(self.tex_infos):
                                kd = self.sample_texture_linear(self.tex_infos[m["DiffuseTexIndex"]], uu, vv)
                            if m["HasAlphaMap"] != 0 and 0 <= m["AlphaTexIndex"] < len(self.tex_infos):
                                alpha = self.sample_mask_linear(self.tex_infos[m["AlphaTexIndex"]], uu, vv)
                            if alpha < m["AlphaCutoff"]:
                                self.cover["cutout_closest"] += 1
                                continue
                            t_closest, n_obj = t, nn
                            if m["TwoSided"] != 0 and dot(n_obj, ray[1]) > 0:
                                self.cover["two_sided_flip"] += 1
                                n_obj = muls(n_obj, f32(-1.0))
                            albedo, tri_out = kd, ti
                    cur = skip
                else:
                    cur = left
            else:
                cur = skip
        return bool(t_closest < f32(1e29)), t_closest, n_obj, albedo, tri_out

    # AnyHit_Tri_Textured :270-327
    def any_hit_tri(self, ray, start, end, t_max):
        cur = start
        while cur != -1 and cur < end:
            bmin, bmax, left, first, count, skip = self.blas[cur]
            if intersect_aabb(ray, bmin, bmax, f32(0.001), t_max):
                if count > 0:
                    for i in range(first, first + count):
                        ti = self.tri_prim[i]
                        v0, v1, v2 = (self.positions[k] for k in self.tris[ti])
                        hit, t, _, bu, bv = intersect_triangle(ray, v0, v1, v2)
                        if not hit:
                            continue
                        if t <= f32(0.001) or t >= t_max:
                            continue
                        m = self.materials[self.tri_mat[ti]]
                        if m["HasAlphaMap"] != 0 and 0 <= m["AlphaTexIndex"] < len(self.tex_infos):
                            uu, vv = self.tri_uv(ti, bu, bv)
                            info = self.tex_infos[m["AlphaTexIndex"]]
                            a_point = self.sample_mask_point(info, uu, vv)
                            cutoff = m["AlphaCutoff"]
                            band = f32(0.10)
                            if a_point < cutoff - band:
                                self.cover["any_point_below"] += 1
                                continue
                            if a_point >= cutoff + band:
                                self.cover["any_point_above"] += 1
                                return True
                            self.cover["any_band_linear"] += 1
                            if self.sample_mask_linear(info, uu, vv) < cutoff:
                                continue
                        return True
                    cur = skip
                else:
                    cur = left
            else:
                cur = skip
        return False

    # TraverseBLAS_Sphere :124-170 -> (hit, tClosest, nObj, albedo, shading, ior)
    def traverse_blas_sphere(self, ray, start, end):
        t_closest, n_obj, albedo, shading, ior = f32(1e30), v3(0, 0, 0), v3(1, 1, 1), 0, f32(1.0)
        cur = start
        while cur != -1 and cur < end:
            bmin, bmax, left, first, count, skip = self.blas[cur]
            if intersect_aabb(ray, bmin, bmax, f32(0.001), t_closest):
                if count > 0:
                    for i in range(first, first + count):
                        prim = self.sphere_prim[i]
                        center, radius, alb, kd, shade, sior, m = self.spheres[prim]
                        hit, t, nn = intersect_sphere(ray, center, radius)
                        if hit and t > f32(0.001) and t < t_closest:
                            t_closest = t
                            n_obj = nn
                            albedo = alb if (kd[0] == 0 and kd[1] == 0 and kd[2] == 0) else kd
                            if m["HasDiffuseMap"] != 0 and 0 <= m["DiffuseTexIndex"] < len(self.tex_infos):        # :149-156 (alpha of the sampler discarded)
                                self.cover["sphere_texture"] += 1
                                u = f32(0.5) + self.math("atan2", nn[2], nn[0]) / (f32(2.0) * PI)
                                v = self.math("acos", fmin(f32(1.0), fmax(f32(-1.0), nn[1]))) / PI
                                albedo = self.sample_texture_linear(self.tex_infos[m["DiffuseTexIndex"]], u, v)
                            shading = shade
                            ior = sior if sior > 0 else f32(1.0)
                    cur = skip
                else:
                    cur = left
            else:
                cur = skip
        return bool(t_closest < f32(1e29)), t_closest, n_obj, albedo, shading, ior

    # AnyHit_Sphere :240-267
    def any_hit_sphere(self, ray, start, end, t_max):
        cur = start
        while cur != -1 and cur < end:
            bmin, bmax, left, first, count, skip = self.blas[cur]
            if intersect_aabb(ray, bmin, bmax, f32(0.001), t_max):
                if count > 0:
                    for i in range(first, first + count):
                        center, radius = self.spheres[self.sphere_prim[i]][:2]
                        hit, t, _ = intersect_sphere(ray, center, radius)
                        if hit and t > f32(0.001) and t < t_max:
                            return True
                    cur = skip
                else:
                    cur = left
            else:
                cur = skip
        return False

    # TraceClosest :30-86 -> (hit, t, normal, albedo, objId, shade, ior)
    def trace_closest(self, wray):
        closest, best_n, best_alb, best_obj, best_shade, best_ior = f32(1e30), v3(0, 0, 0), v3(1, 1, 1), -1, 0, f32(1.0)
        cur = 0
        while cur != -1:
            bmin, bmax, left, first, count, skip = self.tlas[cur]
            if intersect_aabb(wray, bmin, bmax, f32(0.001), closest):
                if count > 0:
                    for i in range(first, first + count):
                        root, ncount, o2w, w2o, uscale, itype = self.inst[self.tlas_inst[i]]
                        iray = transform_ray(w2o, wray)
                        scale = uscale if uscale > 0 else f32(1.0)
                        if itype == 1:                                               # BlasType.SphereSet
                            hit, t_obj, n_obj, albedo, shade, ior = self.traverse_blas_sphere(iray, root, root + ncount)
                            tri_local = -1
                        else:
                            hit, t_obj, n_obj, albedo, tri_local = self.traverse_blas_tri(iray, root, root + ncount)
                            shade, ior = 0, f32(1.0)
                        if hit:
                            t_world = t_obj / scale
                            if t_world < closest:
                                closest = t_world
                                best_n = normalize(transform_vector(o2w, n_obj))
                                best_alb, best_obj, best_shade, best_ior = albedo, tri_local, shade, ior
                    cur = skip
                else:
                    cur = left
            else:
                cur = skip
        return bool(closest < f32(1e29)), closest, best_n, best_alb, best_obj, best_shade, best_ior

    # ShadowOcclusion :89-121
    def shadow_occlusion(self, sray, t_max_world):
        cur = 0
        while cur != -1:
            bmin, bmax, left, first, count, skip = self.tlas[cur]
            if intersect_aabb(sray, bmin, bmax, f32(0.001), t_max_world):
                if count > 0:
                    for i in range(first, first + count):
                        root, ncount, _, w2o, uscale, itype = self.inst[self.tlas_inst[i]]
                        oray = transform_ray(w2o, sray)
                        scale = uscale if uscale > 0 else f32(1.0)
                        t_max_obj = t_max_world * scale
                        blocked = self.any_hit_sphere(oray, root, root + ncount, t_max_obj) if itype == 1 else self.any_hit_tri(oray, root, root + ncount, t_max_obj)
                        if blocked:
                            return True
                    cur = skip
                else:
                    cur = left
            else:
                cur = skip
        return False


# ---------------------------------------------------------------- shading helpers (RTRay.cs:560-655)
def reflect(i, n): return sub(i, muls(n, f32(2.0) * dot(i, n)))                    # :561


def refract(i, n, eta_i, eta_t):                                                    # :564-572 -> (ok, T)
    eta = eta_i / eta_t
    cos_i = -dot(i, n)
    k = f32(1.0) - eta * eta * (f32(1.0) - cos_i * cos_i)
    if k < 0:
        return False, v3(0, 0, 0)
    return True, normalize(add(muls(i, eta), muls(n, eta * cos_i - np.sqrt(k))))


def schlick_fresnel(cos, eta_i, eta_t):                                             # :575-583
    r0 = (eta_i - eta_t) / (eta_i + eta_t)
    r0 = r0 * r0
    om = f32(1.0) - cos
    om2 = om * om
    om5 = om2 * om2 * om
    return r0 + (f32(1.0) - r0) * om5


def orthonormal_basis(n):                                                           # :601-606
    up = v3(0, 1, 0) if abs(n[1]) < f32(0.999) else v3(1, 0, 0)
    t = normalize(cross(up, n))
    return t, cross(n, t)


def sample_hemisphere_cosine(n, rng, sincos):                                       # :586-598
    r1 = rng.next_float(); r2 = rng.next_float()
    phi = f32(2.0) * PI * r1
    cos_theta = np.sqrt(f32(1.0) - r2)
    sin_theta = np.sqrt(r2)
    sn, cs = sincos(phi)
    x = cs * sin_theta
    y = sn * sin_theta
    z = cos_theta
    t, b = orthonormal_basis(n)
    return normalize(add(add(muls(t, x), muls(b, y)), muls(n, z)))


def luminance(c): return f32(0.2126) * c[0] + f32(0.7152) * c[1] + f32(0.0722) * c[2]   # :627
def cos_hemisphere_pdf(n, wi): return fmax(f32(0), dot(n, wi)) * INV_PI                 # :630-634


def safe_color(c):                                                                  # :646-655
    out = []
    for v in c:
        x = v if np.isfinite(v) else f32(0)
        out.append(fmin(f32(1e6), fmax(f32(-1e6), x)))
    return tuple(out)


def float_to_i16(x):                                                                # :609-613
    cl = fmax(f32(0), fmin(f32(65535), x * f32(1000)))
    return trunc_int(cl) & 0xFFFF


def to_i32(v):
    v &= U32
    return v - (1 << 32) if v >= (1 << 31) else v


def to_byte(x):                                                                     # :72-76
    c = fmin(f32(1), fmax(f32(0), x))
    return trunc_int(f32(255.99) * c)


def pack_rgba8(c):                                                                  # :66-70
    v = (255 << 24) | (to_byte(c[0]) << 16) | (to_byte(c[1]) << 8) | to_byte(c[2])
    return v - (1 << 32) if v >= (1 << 31) else v


class Frame:
    """The scalar fields of GBufferParams / IntegratorParams (RTRay.cs:112-146) from the ctypes FrameParams."""

    def __init__(self, p):
        cam = lambda c: {k: (f32(getattr(c, k).X), f32(getattr(c, k).Y), f32(getattr(c, k).Z)) for k in ("origin", "lowerLeft", "horizontal", "vertical")}
        t3 = lambda v: (f32(v.X), f32(v.Y), f32(v.Z))
        self.width, self.height, self.frame = p.width, p.height, p.frame
        self.cam = cam(p.cam)
        self.dir_light_dir, self.dir_light_radiance = t3(p.dirLightDir), t3(p.dirLightRadiance)
        self.sky_top, self.sky_bottom = t3(p.skyTintTop), t3(p.skyTintBottom)
        self.lock, self.spp, self.max_depth = p.rngLockNoise, p.spp, p.maxDepth
        self.temporal, self.spatial = p.enableTemporalReuse, p.enableSpatialReuse
        pc = p.prevCam
        self.prev_cam = dict(origin=t3(pc.origin), right=t3(pc.right), up=t3(pc.up), forward=t3(pc.forward), aspect=f32(pc.aspect), fovY=f32(pc.fovYRadians))
        self.gb = None                                                                  # launch 1's output, read by SpatialCompatible
        self.prev = None                                                                # resPrev (dict of arrays) or None

    def primary_ray(self, index):                                                   # :120-126
        x, y = index % self.width, index // self.width
        u = (f32(x) + f32(0.5)) / f32(max(1, self.width))
        v = (f32(y) + f32(0.5)) / f32(max(1, self.height))
        return generate_ray(self.cam, u, v)

    def sky(self, d):                                                               # SkyWeighted :164-168
        tbg = f32(0.5) * (d[1] + f32(1.0))
        return add(muls(self.sky_bottom, f32(1.0) - tbg), muls(self.sky_top, tbg))

    def distance_from_camera(self, p):                                              # :158-162
        d = sub(p, self.cam["origin"])
        return np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])


def render(arrs, params, math, prev=None, cur=None):
    """PrimaryVisibilityKernel (RTRay.cs:188-201) then PathTraceKernel (:203-325) over the whole image.
    math(name, x[, y]) -> float32 of the shared arithmetic contract ("sin", "cos", "tan", "atan2", "acos"); prev = the previous
    frame's reservoir arrays (res_*), read only; cur = this frame's reservoir arrays, written only where a sample has a diffuse vertex (a
    pixel without one keeps what the buffer held, as resCur does in the reference).  Returns a dict of output arrays named like hrt_outputs."""
    V, K = Views(arrs), Frame(params)
    V.math = K.math = math
    K.prev = prev
    sincos = lambda phi: (math("sin", phi), math("cos", phi))
    n_pix = K.width * K.height
    out = {"color": np.zeros(n_pix, np.int32), "depth": np.zeros(n_pix, np.float32), "objectId": np.zeros(n_pix, np.int32),
           "radiance": np.zeros((n_pix, 3), np.float32), "gb_worldPos": np.zeros((n_pix, 3), np.float32), "gb_normalWS": np.zeros((n_pix, 3), np.float32),
           "gb_baseColor": np.zeros((n_pix, 3), np.float32), "gb_matId": np.zeros(n_pix, np.int32), "gb_objId": np.zeros(n_pix, np.int32),
           "gb_hitMask": np.zeros(n_pix, np.int32), "res_L": np.zeros((n_pix, 3), np.float32), "res_wi": np.zeros((n_pix, 3), np.float32),
           "res_pdf": np.zeros(n_pix, np.float32), "res_w": np.zeros(n_pix, np.float32), "res_wSum": np.zeros(n_pix, np.float32),
           "res_m": np.zeros(n_pix, np.int32), "res_lightId": np.zeros(n_pix, np.int32)}
    if cur is not None:
        out.update({k: cur[k] for k in out if k.startswith("res_")})
    with np.errstate(all="ignore"):
        gb = []
        for index in range(n_pix):                                                  # launch 1
            wray = K.primary_ray(index)
            hit, t, n, alb, obj, shade, ior = V.trace_closest(wray)
            if not hit:                                                             # StoreMiss :100-108
                g = (0, add(wray[0], muls(wray[1], f32(1e6))), v3(0, 1, 0), v3(0, 0, 0), -1, -1)
            else:                                                                   # StoreHit :90-98
                g = (1, add(wray[0], muls(wray[1], t)), n, alb, to_i32((shade & 0xFFFF) | (float_to_i16(ior) << 16)), obj)     # (int arithmetic wraps: ior >= 32.768)
            gb.append(g)
            K.gb = gb
            out["gb_hitMask"][index], out["gb_worldPos"][index], out["gb_normalWS"][index] = g[0], g[1], g[2]
            out["gb_baseColor"][index], out["gb_matId"][index], out["gb_objId"][index] = g[3], g[4], g[5]
        for index in range(n_pix):                                                  # launch 2
            lframe = v3(0, 0, 0)
            hit_mask, g_pos, g_nrm, g_alb, g_mat, g_obj = gb[index]
            for s in range(max(1, K.spp)):
                rng = rng_from_index(index, K.width, K.height, K.frame, s, 0xC0FFEE, K.lock)
                if hit_mask == 0:
                    lframe = add(lframe, safe_color(K.sky(K.primary_ray(index)[1])))
                    continue
                pos, nrm, alb = g_pos, normalize(g_nrm), g_alb
                shade = g_mat & 0xFFFF
                ior = f32((g_mat >> 16) & 0xFFFF) / f32(1000.0)                     # I16ToFloat :615
                li, thr = v3(0, 0, 0), v3(1, 1, 1)
                i_dir = normalize(sub(pos, K.cam["origin"]))                         # ViewDirFromCam :156
                wrote = False
                depth = 0
                while depth < K.max_depth:
                    if shade == SHADING_MIRROR:                                      # :235-244
                        ray = make_ray_with_normal_offset(pos, nrm, reflect(i_dir, nrm))
                        thr = mulv(thr, alb)
                    elif shade == SHADING_GLASS:                                     # :246-275
                        nuse = nrm
                        outside = bool(dot(i_dir, nrm) < 0)
                        if not outside:
                            nuse = muls(nuse, f32(-1.0))
                        eta_i = f32(1.0) if outside else (ior if ior > 0 else f32(1.5))
                        eta_t = (ior if ior > 0 else f32(1.5)) if outside else f32(1.0)
                        dir_r = reflect(i_dir, nuse)
                        refr_ok, dir_t = refract(i_dir, nuse, eta_i, eta_t)
                        fr = schlick_fresnel(abs(dot(i_dir, nuse)), eta_i, eta_t)
                        xi = rng.next_float()
                        if (not refr_ok) or xi < fr:
                            ray = make_ray_with_normal_offset(pos, nuse, dir_r)
                        else:
                            ray = make_ray_with_normal_offset(pos, neg(nuse), dir_t)
                        if refr_ok and xi >= fr:
                            tint = v3(1, 1, 1) if (alb[0] == 0 and alb[1] == 0 and alb[2] == 0) else alb
                            thr = muls(mulv(thr, tint), (eta_i * eta_i) / (eta_t * eta_t))
                    else:                                                            # :277-317
                        direct, res = restir_direct(V, K, index, pos, nrm, alb, rng, sincos, reuse=not wrote)   # kLocal :280-285
                        li = add(li, mulv(thr, direct))
                        if not wrote:                                                # resCur.Write :42-47, every sample's first diffuse vertex
                            out["res_L"][index], out["res_wi"][index] = res["L"], res["wi"]
                            out["res_pdf"][index], out["res_w"][index], out["res_wSum"][index] = res["pdf"], res["w"], res["wSum"]
                            out["res_lightId"][index], out["res_m"][index] = res["lightId"], res["m"]
                            wrote = True
                        wi = sample_hemisphere_cosine(nrm, rng, sincos)
                        ray = make_ray_with_normal_offset(pos, nrm, wi)
                        thr = mulv(thr, alb)
                        if depth >= 3:                                               # :306-311
                            max_c = fmax(thr[0], fmax(thr[1], thr[2]))
                            max_c = fmax(f32(0.05), fmin(f32(0.98), max_c))          # XMath.Clamp
                            if rng.next_float() > max_c:
                                thr = v3(0, 0, 0)
                                break
                            thr = muls(thr, f32(1.0) / max_c)
                    hit, t, n2, alb2, _, shade2, ior2 = V.trace_closest(ray)         # TraceNext :659-671
                    if not hit:
                        li = add(li, mulv(thr, K.sky(ray[1])))
                        break
                    pos, nrm, alb, shade, ior = add(ray[0], muls(ray[1], t)), normalize(n2), alb2, shade2, ior2
                    i_dir = ray[1]
                    depth += 1
                lframe = add(lframe, safe_color(li))
            lout = muls(lframe, f32(1.0) / f32(max(1, K.spp)))
            out["radiance"][index] = lout
            out["color"][index] = pack_rgba8(lout)
            out["depth"][index] = K.distance_from_camera(g_pos)
            out["objectId"][index] = g_obj
    out["_cover"] = V.cover
    return out


def reservoir_update(r, wi, pdf_sel, li, score, light_id, rng):                     # :394-405 (multiplicity 1)
    new_sum = r["wSum"] + score
    accept = score / new_sum if new_sum > 0 else f32(0)
    if rng.next_float() < accept:
        r["wi"], r["pdf"], r["L"], r["w"], r["lightId"] = wi, pdf_sel, li, score, light_id
    r["wSum"] = new_sum
    r["m"] = r["m"] + 1


def reproject_to_prev_pixel(K, pos):                                                # :339-360
    c = K.prev_cam
    p = sub(pos, c["origin"])
    x, y, z = dot(p, c["right"]), dot(p, c["up"]), dot(p, c["forward"])
    if z <= f32(1e-4):
        return -1
    tan_half = K.math("tan", f32(0.5) * c["fovY"])
    ndc_x = x / (z * tan_half * c["aspect"])
    ndc_y = y / (z * tan_half)
    fx = f32(0.5) * (ndc_x + f32(1.0)) * f32(K.width)
    fy = f32(0.5) * (ndc_y + f32(1.0)) * f32(K.height)
    if not (np.isfinite(fx) and np.isfinite(fy)):                                    # (int)NaN / (int)inf = int.MinValue: outside as unsigned
        return -1
    px, py = trunc_int(fx), trunc_int(fy)
    if not (0 <= px < K.width and 0 <= py < K.height):
        return -1
    return py * K.width + px


def spatial_compatible(K, a, b, n_a):                                               # :363-374
    if K.gb[a][5] == K.gb[b][5]:
        return True
    if dot(n_a, normalize(K.gb[b][2])) < f32(0.85):
        return False
    z_a, z_b = K.distance_from_camera(K.gb[a][1]), K.distance_from_camera(K.gb[b][1])
    return bool(abs(z_a - z_b) / fmax(f32(1e-3), z_a) < f32(0.05))


def neighbor8(rot, r):                                                              # :377-391
    rx = lambda x, y: x if rot == 0 else (-y if rot == 1 else (-x if rot == 2 else y))
    ry = lambda x, y: y if rot == 0 else (x if rot == 1 else (-y if rot == 2 else -x))
    return [(rx(x, y), ry(x, y)) for x, y in ((-r, 0), (r, 0), (0, -r), (0, r), (-r, -r), (r, -r), (-r, r), (r, r))]


def import_from_prev(K, prev_idx, cur_idx, n, albedo, mix_local, mix_delta, rng, r):  # :408-435
    P = K.prev
    if prev_idx < 0 or P is None or len(P["res_L"]) <= prev_idx:
        return
    if not spatial_compatible(K, cur_idx, prev_idx, n):
        return
    pm, pw, pw_sum = int(P["res_m"][prev_idx]), f32(P["res_w"][prev_idx]), f32(P["res_wSum"][prev_idx])
    if not (pm > 0 and pw > 0 and pw_sum > 0):
        return
    wi = tuple(f32(v) for v in P["res_wi"][prev_idx])
    lid = 2 if int(P["res_lightId"][prev_idx]) == 2 else 1
    li_imp = K.dir_light_radiance if lid == 2 else K.sky(wi)
    nl = fmax(f32(0), dot(n, wi))
    pdf_here = fmax(EPS_MIN, mix_delta) if lid == 2 else fmax(EPS_MIN, cos_hemisphere_pdf(n, wi) * mix_local)
    s_here = luminance(muls(mulv(albedo, li_imp), (nl / pdf_here) * INV_PI))
    w_src = pw_sum / (f32(max(1, pm)) * fmax(EPS_MIN, pw))
    reservoir_update(r, wi, pdf_here, li_imp, s_here * w_src, lid, rng)


def hash3(a, b, c): return _hash32(a ^ _hash32(b ^ _hash32(c)))                     # :643


def restir_direct(V, K, index, pos, n, albedo, rng, sincos, reuse):                  # :438-543
    mix_local = f32(8) / f32(9)
    mix_delta = f32(1) / f32(9)
    r = {"L": v3(0, 0, 0), "wi": v3(0, 0, 0), "pdf": f32(0), "w": f32(0), "wSum": f32(0), "m": 0, "lightId": 0}
    for _ in range(8):                                                               # (1) :449-460
        wi = sample_hemisphere_cosine(n, rng, sincos)
        nl = fmax(f32(0), dot(n, wi))
        pdf_local = fmax(EPS_MIN, cos_hemisphere_pdf(n, wi))
        pdf_sel = fmax(EPS_MIN, pdf_local * mix_local)
        li_loc = K.sky(wi)
        f_over_p = muls(mulv(albedo, li_loc), (nl / pdf_sel) * INV_PI)
        reservoir_update(r, wi, pdf_sel, li_loc, luminance(f_over_p), 1, rng)
    wi = normalize(K.dir_light_dir)                                                  # (2) :463-471
    nl = fmax(f32(0), dot(n, wi))
    pdf_sel = fmax(EPS_MIN, mix_delta)
    f_over_p = muls(mulv(albedo, K.dir_light_radiance), (nl / pdf_sel) * INV_PI)
    reservoir_update(r, wi, pdf_sel, K.dir_light_radiance, luminance(f_over_p), 2, rng)
    if reuse and K.temporal != 0:                                                    # (3) :474-481
        prev_idx = reproject_to_prev_pixel(K, pos)
        if prev_idx >= 0:
            import_from_prev(K, prev_idx, index, n, albedo, mix_local, mix_delta, rng, r)
    if reuse and K.spatial != 0:                                                     # (4) :484-514
        h = hash3(index & U32, K.frame & U32, 0xB31F5AB1)
        rot, radius = h & 3, 1 + ((h >> 2) & 1)
        x0, y0 = index % K.width, index // K.width
        nbr = [((y0 + dy) * K.width + (x0 + dx)) if (0 <= x0 + dx < K.width and 0 <= y0 + dy < K.height) else -1 for dx, dy in neighbor8(rot, radius)]
        for q in nbr:
            import_from_prev(K, q, index, n, albedo, mix_local, mix_delta, rng, r)
    contrib = v3(0, 0, 0)                                                            # (5) :518-539
    if r["m"] > 0 and r["wSum"] > 0 and r["w"] > 0:
        wi_sel = r["wi"]
        lid = 2 if r["lightId"] == 2 else 1
        nl_sel = fmax(f32(0), dot(n, wi_sel))
        visible = False
        if nl_sel > 0 and dot(n, wi_sel) > 0:                                        # Visible :617-624
            visible = not V.shadow_occlusion(make_ray_with_normal_offset(pos, n, wi_sel), f32(1e29))
        if visible:
            pdf_sel = fmax(EPS_MIN, mix_delta) if lid == 2 else fmax(EPS_MIN, cos_hemisphere_pdf(n, wi_sel) * mix_local)
            li_sel = K.dir_light_radiance if lid == 2 else K.sky(wi_sel)
            f_over_p = muls(mulv(albedo, li_sel), (nl_sel / pdf_sel) * INV_PI)
            w = r["wSum"] / f32(max(1, r["m"])) / fmax(EPS_MIN, r["w"])
            contrib = muls(f_over_p, w)
    return contrib, r
