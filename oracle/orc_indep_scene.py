"""oracle/orc_indep_scene.py -- a SECOND, independent restatement of the reference's host-side scene builders.  TEST INFRASTRUCTURE ONLY.

oracle/orc_scene.hpp (C++) and the library's csrc/hrt_host.cpp both restate Engine/Scene.cs; tests/test_host_scene.py holds them to
byte-identical arrays -- which pins them to each other, not to the reference.  This file reads Scene.cs again (paths under
/root/reference/ILGPU_Raytracing/Engine/, line numbers below), as scalar numpy.float32 Python in the reference's statement order,
with the .NET semantics of what it calls on the HOST: Math.Min / Math.Max behind XMath.Min / Max on a CPU (a NaN operand gives NaN,
-0 < +0), Array.Sort = ArraySortHelper<T>.IntrospectiveSort of .NET 8 (unstable: decides the topology under equal keys), List<T>
append order, and the two position-indexed look-ups (BuildBLAS_Spheres :381-395, BuildBLAS_Triangles :398-403).
tests/test_oracle_indep.py requires it to produce the C++ oracle's arrays byte for byte on the fuzz recipes and the hostile scenes.
"""
import math

import numpy as np

from ilgpu_raytracing_amd import _types as T

f32 = np.float32
FMAX = f32(np.finfo(np.float32).max)          # float.MaxValue; float.MinValue = -FMAX


# ---------------------------------------------------------------- .NET Math.Min / Math.Max on floats (XMath.Min / Max on a CPU)
def nmin(a, b):
    if a != a: return a
    if b != b: return b
    if a == b: return a if np.signbit(a) else b           # -0 is smaller than +0
    return a if a < b else b


def nmax(a, b):
    if a != a: return a
    if b != b: return b
    if a == b: return b if np.signbit(a) else a
    return a if a > b else b


def v3(x, y, z): return (f32(x), f32(y), f32(z))
def vmin(a, b): return (nmin(a[0], b[0]), nmin(a[1], b[1]), nmin(a[2], b[2]))          # Float3.cs:67-70
def vmax(a, b): return (nmax(a[0], b[0]), nmax(a[1], b[1]), nmax(a[2], b[2]))          # Float3.cs:73-76
def f3_of(rec): return (f32(rec.X), f32(rec.Y), f32(rec.Z))


def normalize(v):                                                                        # Float3.cs:91-95 (Rsqrt on a CPU = 1 / Sqrt)
    with np.errstate(all="ignore"):
        inv = f32(1.0) / np.sqrt(nmax(f32(1e-20), v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
        return (v[0] * inv, v[1] * inv, v[2] * inv)


def length(v):                                                                           # Float3.cs:104-107
    with np.errstate(all="ignore"):
        return np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


# ---------------------------------------------------------------- Array.Sort(idx, start, count, comparer): .NET 8 IntrospectiveSort
def dotnet_sort(k, lo0, n0, cmp):
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
    if n0 >= 2:
        intro(lo0, n0, 2 * (int(math.floor(math.log2(n0))) + 1))


def by_key(key):
    def cmp(a, b):
        ka, kb = key(a), key(b)
        return -1 if ka < kb else (1 if ka > kb else 0)
    return cmp


# ---------------------------------------------------------------- transforms (Scene.cs:560-580, 616-656)
def transform_point(m, p):
    with np.errstate(all="ignore"):
        return tuple(m[r][0] * p[0] + m[r][1] * p[1] + m[r][2] * p[2] + m[r][3] for r in range(3))


def transform_vector(m, v):
    with np.errstate(all="ignore"):
        return tuple(m[r][0] * v[0] + m[r][1] * v[1] + m[r][2] * v[2] for r in range(3))


def transform_aabb(m, bmin, bmax):
    c = [(bmin[0], bmin[1], bmin[2]), (bmax[0], bmin[1], bmin[2]), (bmin[0], bmax[1], bmin[2]), (bmin[0], bmin[1], bmax[2]),
         (bmax[0], bmax[1], bmin[2]), (bmin[0], bmax[1], bmax[2]), (bmax[0], bmin[1], bmax[2]), (bmax[0], bmax[1], bmax[2])]
    mn, mx = (FMAX, FMAX, FMAX), (-FMAX, -FMAX, -FMAX)
    for p in c:
        w = transform_point(m, p)
        mn, mx = vmin(mn, w), vmax(mx, w)
    return mn, mx


def invert_rigid_or_uniform(m):
    with np.errstate(all="ignore"):
        col = lambda c: (m[0][c], m[1][c], m[2][c])
        sx, sy, sz = length(col(0)), length(col(1)), length(col(2))
        uni = (sx + sy + sz) / f32(3.0)
        inv = f32(1.0) / uni if uni > 0 else f32(1.0)
        r0, r1, r2 = normalize(col(0)), normalize(col(1)), normalize(col(2))
        im = [[r0[0] * inv, r1[0] * inv, r2[0] * inv, f32(0)], [r0[1] * inv, r1[1] * inv, r2[1] * inv, f32(0)], [r0[2] * inv, r1[2] * inv, r2[2] * inv, f32(0)]]
        it = transform_vector(im, (m[0][3], m[1][3], m[2][3]))
        it = (it[0] * f32(-1.0), it[1] * f32(-1.0), it[2] * f32(-1.0))
        im[0][3], im[1][3], im[2][3] = it
    return im, uni


def affine_rows(a):
    return [[f32(getattr(a, "m%d%d" % (r, c))) for c in range(4)] for r in range(3)]


# ---------------------------------------------------------------- the scene (host lists of Scene.cs:33-60)
class IndepScene:
    """add_sphere / build_sphere_instance / load_mesh_instance / rebuild_tlas with the argument conventions of the other two builders."""

    def __init__(self):
        self.spheres, self.sphere_prim = [], []
        self.blas = []                                  # dicts: bmin, bmax, left, right, first, count, skip
        self.instances = []
        self.tri_prim, self.positions, self.tris, self.texcoords, self.tri_uvs, self.tri_mat = [], [], [], [], [], []
        self.materials, self.texels, self.tex_infos = [], [], []
        self.tlas, self.tlas_idx = [], []

    # AddSphere :315-321
    def add_sphere(self, s):
        self.spheres.append(T.Sphere.from_buffer_copy(s))
        self.sphere_prim.append(len(self.spheres) - 1)
        return len(self.spheres) - 1

    def add_texture(self, bgra_or_rgba):                # (not a Scene.cs method: the other builders' helper for sphere textures)
        t = np.ascontiguousarray(bgra_or_rgba, np.uint8)
        self.tex_infos.append((len(self.texels), t.shape[1], t.shape[0]))
        self.texels.extend(tuple(int(v) for v in px) for px in t.reshape(-1, 4))
        return len(self.tex_infos) - 1

    def _sphere_c_r(self, sid):
        s = self.spheres[sid]
        return f3_of(s.center), f32(s.radius)

    # BuildBLASNodeRecursive :405-468
    def _build_blas(self, prim_idx, idx, start, count, bmin_pre, bmax_pre, parent_skip, spheres, tri_bounds, tri_center):
        node_index = len(self.blas)
        nb_min, nb_max = (FMAX, FMAX, FMAX), (-FMAX, -FMAX, -FMAX)
        if bmin_pre is not None:
            for i in range(start, start + count):
                nb_min, nb_max = vmin(nb_min, bmin_pre[i]), vmax(nb_max, bmax_pre[i])     # (sic: indexed by POSITION in idx, :415-418)
        else:
            for i in range(start, start + count):
                mn, mx = tri_bounds(prim_idx[idx[i]])
                nb_min, nb_max = vmin(nb_min, mn), vmax(nb_max, mx)
        node = dict(bmin=nb_min, bmax=nb_max, left=-1, right=-1, first=-1, count=0, skip=parent_skip)
        self.blas.append(node)
        if count <= 4:
            leaf_start = len(prim_idx)
            for i in range(start, start + count):
                prim_idx.append(prim_idx[idx[i]])
            node.update(first=leaf_start, count=count, skip=parent_skip)
            return node_index
        with np.errstate(all="ignore"):
            ext = (nb_max[0] - nb_min[0], nb_max[1] - nb_min[1], nb_max[2] - nb_min[2])
        axis = 0
        if ext[1] > ext[0] and ext[1] >= ext[2]: axis = 1
        elif ext[2] > ext[0] and ext[2] >= ext[1]: axis = 2
        if spheres:
            key = lambda a: f3_of(self.spheres[prim_idx[a]].center)[axis]
        else:
            key = lambda a: tri_center(prim_idx[a])[axis]
        dotnet_sort(idx, start, count, by_key(key))
        mid = start + (count >> 1)
        right_root = self._build_blas(prim_idx, idx, mid, count - (mid - start), bmin_pre, bmax_pre, parent_skip, spheres, tri_bounds, tri_center)
        left_root = self._build_blas(prim_idx, idx, start, mid - start, bmin_pre, bmax_pre, right_root, spheres, tri_bounds, tri_center)
        node.update(left=left_root, right=right_root, skip=parent_skip)
        return node_index

    # BuildSphereInstance :323-356 + BuildBLAS_Spheres :381-395
    def build_sphere_instance(self, sphere_ids, o2w=None):
        ids = list(sphere_ids)
        m = affine_rows(o2w if o2w is not None else T.identity_affine())
        bmin, bmax = (FMAX, FMAX, FMAX), (-FMAX, -FMAX, -FMAX)
        with np.errstate(all="ignore"):
            for sid in ids:
                c, r = self._sphere_c_r(sid)
                bmin = vmin(bmin, (c[0] - r, c[1] - r, c[2] - r))
                bmax = vmax(bmax, (c[0] + r, c[1] + r, c[2] + r))
            prim_start, prim_count = ids[0], len(ids)
            blas_start = len(self.blas)
            idx = [prim_start + i for i in range(prim_count)]
            pre_min, pre_max = [], []
            for i in range(prim_count):
                c, r = self._sphere_c_r(self.sphere_prim[prim_start + i])                  # by POSITION in the prim-index list (:386)
                pre_min.append((c[0] - r, c[1] - r, c[2] - r)); pre_max.append((c[0] + r, c[1] + r, c[2] + r))
        self._build_blas(self.sphere_prim, idx, 0, prim_count, pre_min, pre_max, -1, True, None, None)
        wmin, wmax = transform_aabb(m, bmin, bmax)
        w2o, uni = invert_rigid_or_uniform(m)
        self.instances.append(dict(type=1, root=blas_start, count=len(self.blas) - blas_start, first=prim_start, n=prim_count, o2w=m, w2o=w2o, uni=uni, wmin=wmin, wmax=wmax))
        return len(self.instances) - 1

    # LoadMeshInstance :143-256 (after MeshLoaderOBJ or a caller built the MeshHost)
    def load_mesh_instance(self, mesh, o2w=None):
        m = affine_rows(o2w if o2w is not None else T.identity_affine())
        base_v, base_t, base_uv, base_mat = len(self.positions), len(self.tris), len(self.texcoords), len(self.materials)
        own_pos = [v3(*p) for p in mesh.positions]
        self.positions.extend(own_pos)
        self.texcoords.extend((f32(t[0]), f32(t[1])) for t in mesh.texcoords)
        n_tri = len(mesh.triangles)
        tri_mat = mesh.tri_mat if mesh.tri_mat is not None else []
        for i in range(n_tri):
            t = mesh.triangles[i]
            self.tris.append((int(t[0]) + base_v, int(t[1]) + base_v, int(t[2]) + base_v))
            u = mesh.tri_uvs[i]
            self.tri_uvs.append((int(u[0]) + base_uv, int(u[1]) + base_uv, int(u[2]) + base_uv))
            self.tri_mat.append(base_mat + (int(tri_mat[i]) if i < len(tri_mat) else 0))
            self.tri_prim.append(base_t + i)
        n_tex = mesh.n_tex
        tex_arrays, off = [], 0
        for k in range(n_tex):
            n = int(mesh.tex_w[k]) * int(mesh.tex_h[k]) * 4
            tex_arrays.append(np.asarray(mesh.tex_bytes[off:off + n], np.uint8).reshape(-1, 4)); off += n

        def append_tex(k):
            start = len(self.texels)
            self.texels.extend((int(p[2]), int(p[1]), int(p[0]), int(p[3])) for p in tex_arrays[k])       # BGRA -> R, G, B, A (:190-191)
            self.tex_infos.append((start, int(mesh.tex_w[k]), int(mesh.tex_h[k])))
            return len(self.tex_infos) - 1
        for i in range(mesh.n_materials):
            mr = T.MaterialRecord.from_buffer_copy(mesh.materials[i])
            if mr.HasDiffuseMap != 0 and 0 <= mr.DiffuseTexIndex < n_tex:
                mr.DiffuseTexIndex = append_tex(mr.DiffuseTexIndex); mr.HasDiffuseMap = 1
            else:
                mr.HasDiffuseMap = 0; mr.DiffuseTexIndex = -1
            if mr.HasAlphaMap != 0 and 0 <= mr.AlphaTexIndex < n_tex:
                mr.AlphaTexIndex = append_tex(mr.AlphaTexIndex); mr.HasAlphaMap = 1
            else:
                mr.HasAlphaMap = 0; mr.AlphaTexIndex = -1
            self.materials.append(mr)

        def tri_bounds(ti):                                                                 # BoundsOfTriangle :597-605
            a, b, c = (self.positions[k] for k in self.tris[ti])
            return vmin(a, vmin(b, c)), vmax(a, vmax(b, c))

        def tri_center(ti):                                                                 # CenterOfTriangle :607-614
            a, b, c = (self.positions[k] for k in self.tris[ti])
            with np.errstate(all="ignore"):
                return ((a[0] + b[0] + c[0]) / f32(3.0), (a[1] + b[1] + c[1]) / f32(3.0), (a[2] + b[2] + c[2]) / f32(3.0))
        blas_start = len(self.blas)
        idx = [base_t + i for i in range(n_tri)]                                            # looked up BY POSITION in the prim-index list (:400, :428)
        self._build_blas(self.tri_prim, idx, 0, n_tri, None, None, -1, False, tri_bounds, tri_center)
        bmin, bmax = (FMAX, FMAX, FMAX), (-FMAX, -FMAX, -FMAX)                              # ComputeMeshBounds :582-595 over the mesh's OWN lists
        for i in range(n_tri):
            a, b, c = (own_pos[int(k)] for k in mesh.triangles[i])
            bmin, bmax = vmin(bmin, vmin(a, vmin(b, c))), vmax(bmax, vmax(a, vmax(b, c)))
        wmin, wmax = transform_aabb(m, bmin, bmax)
        w2o, uni = invert_rigid_or_uniform(m)
        self.instances.append(dict(type=2, root=blas_start, count=len(self.blas) - blas_start, first=base_t, n=n_tri, o2w=m, w2o=w2o, uni=uni, wmin=wmin, wmax=wmax))
        self.rebuild_tlas()
        return len(self.instances) - 1

    # RebuildTLAS :358-368 + BuildTLASNodeRecursive :470-510
    def rebuild_tlas(self):
        n = len(self.instances)
        idx = list(range(n))
        nodes = []

        def rec(start, count, parent_skip):
            node_index = len(nodes)
            nb_min, nb_max = (FMAX, FMAX, FMAX), (-FMAX, -FMAX, -FMAX)
            for i in range(start, start + count):
                r = self.instances[idx[i]]
                nb_min, nb_max = vmin(nb_min, r["wmin"]), vmax(nb_max, r["wmax"])
            node = dict(bmin=nb_min, bmax=nb_max, left=-1, right=-1, first=-1, count=0, skip=parent_skip)
            nodes.append(node)
            if count <= 2:
                node.update(first=start, count=count, skip=parent_skip)
                return node_index
            with np.errstate(all="ignore"):
                ext = (nb_max[0] - nb_min[0], nb_max[1] - nb_min[1], nb_max[2] - nb_min[2])
                axis = 0
                if ext[1] > ext[0] and ext[1] >= ext[2]: axis = 1
                elif ext[2] > ext[0] and ext[2] >= ext[1]: axis = 2
                key = lambda a: f32(0.5) * (self.instances[a]["wmin"][axis] + self.instances[a]["wmax"][axis])     # Float3.Center :110-113
                dotnet_sort(idx, start, count, by_key(key))
            mid = start + (count >> 1)
            right_root = rec(mid, count - (mid - start), parent_skip)
            left_root = rec(start, mid - start, right_root)
            node.update(left=left_root, right=right_root, skip=parent_skip)
            return node_index
        rec(0, n, -1)                   # (n = 0: one node with the inverted float.MaxValue box and count 0)
        self.tlas, self.tlas_idx = nodes, idx

    # ---------------------------------------------------------------- the 15 arrays, in the wire layout
    def arrays(self):
        def nodes_arr(nodes):
            a = np.zeros(len(nodes), T.np_dtype(T.BvhNode))
            for i, n in enumerate(nodes):
                for k, f in enumerate("XYZ"):
                    a[i]["boundsMin"][f] = n["bmin"][k]; a[i]["boundsMax"][f] = n["bmax"][k]
                a[i]["left"], a[i]["right"], a[i]["first"], a[i]["count"], a[i]["skipIndex"] = n["left"], n["right"], n["first"], n["count"], n["skip"]
            return a
        out = {"tlasNodes": nodes_arr(self.tlas), "tlasInstanceIndices": np.array(self.tlas_idx, np.int32).reshape(-1), "blasNodes": nodes_arr(self.blas)}
        inst = np.zeros(len(self.instances), T.np_dtype(T.InstanceRecord))
        for i, r in enumerate(self.instances):
            inst[i]["type"], inst[i]["blasRoot"], inst[i]["blasNodeCount"], inst[i]["primIndexFirst"], inst[i]["primIndexCount"] = r["type"], r["root"], r["count"], r["first"], r["n"]
            for rr in range(3):
                for cc in range(4):
                    inst[i]["objectToWorld"]["m%d%d" % (rr, cc)] = r["o2w"][rr][cc]; inst[i]["worldToObject"]["m%d%d" % (rr, cc)] = r["w2o"][rr][cc]
            inst[i]["uniformScale"] = r["uni"]
            for k, f in enumerate("XYZ"):
                inst[i]["worldBoundsMin"][f] = r["wmin"][k]; inst[i]["worldBoundsMax"][f] = r["wmax"][k]
        out["instances"] = inst
        out["spherePrimIdx"] = np.array(self.sphere_prim, np.int32).reshape(-1)
        out["spheres"] = np.frombuffer(b"".join(bytes(s) for s in self.spheres), T.np_dtype(T.Sphere)).copy() if self.spheres else np.zeros(0, T.np_dtype(T.Sphere))
        out["triPrimIdx"] = np.array(self.tri_prim, np.int32).reshape(-1)
        pos = np.zeros(len(self.positions), T.np_dtype(T.Float3))
        for i, p in enumerate(self.positions):
            pos[i]["X"], pos[i]["Y"], pos[i]["Z"] = p
        out["meshPositions"] = pos
        out["meshTris"] = np.array([tuple(t) for t in self.tris], T.np_dtype(T.MeshTri)) if self.tris else np.zeros(0, T.np_dtype(T.MeshTri))
        tc = np.zeros(len(self.texcoords), T.np_dtype(T.Float2))
        for i, t in enumerate(self.texcoords):
            tc[i]["X"], tc[i]["Y"] = t
        out["meshTexcoords"] = tc
        out["meshTriUVs"] = np.array([tuple(t) for t in self.tri_uvs], T.np_dtype(T.MeshTriUV)) if self.tri_uvs else np.zeros(0, T.np_dtype(T.MeshTriUV))
        out["triMatIndex"] = np.array(self.tri_mat, np.int32).reshape(-1)
        out["materials"] = np.frombuffer(b"".join(bytes(m) for m in self.materials), T.np_dtype(T.MaterialRecord)).copy() if self.materials else np.zeros(0, T.np_dtype(T.MaterialRecord))
        out["texels"] = np.array(self.texels, T.np_dtype(T.RGBA32)) if self.texels else np.zeros(0, T.np_dtype(T.RGBA32))
        out["texInfos"] = np.array(self.tex_infos, T.np_dtype(T.TexInfo)) if self.tex_infos else np.zeros(0, T.np_dtype(T.TexInfo))
        return out


# ---------------------------------------------------------------- Camera.cs / RTRenderer.cs host code, restated a second time
# `math(name, x)` supplies the shared transcendentals (tan, atan, sin, cos of include/hrt_math.h: XMath's own bits are unknowable
# here); everything else is scalar float32 in the reference's statement order, with the HOST's Max in Normalize.
def _sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def _add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def _mul(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def _cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def _dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def _abs(x): return -x if x < f32(0) else x                                              # Camera.cs Abs: x < 0 ? -x : x (keeps -0 and NaN)
PI = f32(3.14159265358979323846)


def _ortho_basis(forward, up_hint):                                                      # Camera.cs OrthoBasis
    f = normalize(forward)
    up = up_hint
    if _abs(_dot(f, up)) > f32(0.999):
        up = v3(0, 1, 0)
        if _abs(_dot(f, up)) > f32(0.999): up = v3(1, 0, 0)
    u = normalize(_cross(f, up))
    v = normalize(_cross(u, f))
    return u, v, (-f[0], -f[1], -f[2])


def camera_lookat(math, origin, look_at, up, vfov_degrees, aspect, focus_dist=1.0):      # Camera.cs: the six-argument constructor
    with np.errstate(all="ignore"):
        origin, look_at, up = v3(*origin), v3(*look_at), v3(*up)
        aspect, focus_dist = f32(aspect), f32(focus_dist)
        theta = f32(vfov_degrees) * (PI / f32(180))
        half_h = math("tan", f32(0.5) * theta)
        half_w = aspect * half_h
        forward = normalize(_sub(look_at, origin))
        u, v, w = _ortho_basis(forward, up)
        cam = {"origin": origin, "horizontal": _mul(u, f32(2) * half_w), "vertical": _mul(v, f32(2) * half_h)}
        cam["lowerLeft"] = _add(_sub(_sub(origin, _mul(u, half_w)), _mul(v, half_h)), _mul(forward, focus_dist))
        cam["forward"] = normalize(_sub(_add(_add(cam["lowerLeft"], _mul(cam["horizontal"], f32(0.5))), _mul(cam["vertical"], f32(0.5))), origin))
        cam["right"] = normalize(_cross(cam["forward"], v))
        cam["up"] = normalize(v)
        cam["aspect"], cam["fovYRadians"] = aspect, theta
        return cam


def camera_bake(math, cam, pixel_w, pixel_h):                                            # RTRenderer.cs BakeCameraDerived
    with np.errstate(all="ignore"):
        c = dict(cam)
        center = _add(_add(c["lowerLeft"], _mul(c["horizontal"], f32(0.5))), _mul(c["vertical"], f32(0.5)))
        forward = normalize(_sub(center, c["origin"]))
        up = normalize(c["vertical"])
        right = normalize(_cross(forward, up))
        focus = length(_sub(center, c["origin"]))
        half_h = f32(0.5) * length(c["vertical"])
        tan_half = half_h / focus if focus > f32(1e-6) else half_h
        fov_y = f32(2) * math("atan", tan_half)
        if length(c["horizontal"]) > f32(1e-6) and length(c["vertical"]) > f32(1e-6):
            aspect = length(c["horizontal"]) / length(c["vertical"])
        else:
            aspect = f32(pixel_w) / f32(max(1, pixel_h))
        c["forward"], c["up"], c["right"], c["fovYRadians"], c["aspect"] = forward, up, right, fov_y, aspect
        return c


def sun_dir(math, azimuth, elevation):                                                   # RTRenderer.cs:174-178
    with np.errstate(all="ignore"):
        az, el = f32(azimuth), f32(elevation)
        return normalize((math("cos", az) * math("cos", el), math("sin", el), math("sin", az) * math("cos", el)))
