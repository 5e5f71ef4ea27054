"""Synthetic scenes of the five BASELINE.json configs.

None of them exists in the reference (its only built-in scene is the 6-sphere default
scene); each is expressed through the reference's own scene-build API -- AddSphere,
BuildSphereInstance (one sphere per instance, the only multi-sphere layout the
reference builds correctly), LoadObjInstance's array path, RebuildTLAS --
so the same calls drive the product's host builder (`engine.Scene`) and the oracle's
(`oracle.orc.OrcScene`): `build(cfg, builder)` only needs those method names.

Data is deterministic: xorshift32 (`RNG.NextFloat`, RTUtils.cs:33-49) for config 3,
closed-form displacement for the meshes.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import _types as T
from .engine import MeshData


@dataclass
class Config:
    name: str
    width: int
    height: int
    spp: int
    cam_origin: tuple
    cam_lookat: tuple
    max_depth: int = 3
    vfov: float = 60.0
    description: str = ""
    extra: dict = field(default_factory=dict)


def material(kd=(1.0, 1.0, 1.0), diffuse_tex=-1, alpha_tex=-1, two_sided=0, alpha_cutoff=0.5, shading=0, ior=1.0):
    m = T.MaterialRecord()
    m.Kd = T.f3(*kd)
    m.HasDiffuseMap = 1 if diffuse_tex >= 0 else 0
    m.DiffuseTexIndex = diffuse_tex
    m.Shading = shading
    m.IOR = ior
    m.HasAlphaMap = 1 if alpha_tex >= 0 else 0
    m.AlphaTexIndex = alpha_tex
    m.TwoSided = two_sided
    m.AlphaCutoff = alpha_cutoff
    return m


def sphere(center, radius, albedo, shading=T.SHADING_LAMBERT, ior=1.0, mat=None):
    s = T.Sphere()
    s.center = T.f3(*center)
    s.radius = float(radius)
    s.albedo = T.f3(*albedo)
    s.material = mat if mat is not None else material(kd=albedo)
    s.shading = shading
    s.ior = float(ior)
    return s


class XorShift32:
    """RNG.Create / NextUInt / NextFloat (RTUtils.cs:25-49) in float32 arithmetic."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF or 1

    def next_u(self):
        x = self.s
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        self.s = x if x != 0 else 1
        return self.s

    def next_f(self):
        return np.float32(self.next_u() & 0x00FFFFFF) * np.float32(1.0 / 16777216.0)

    def uniform(self, lo, hi):
        return float(np.float32(lo) + (np.float32(hi) - np.float32(lo)) * self.next_f())


CONFIGS = {
    1: Config("config1_single_sphere", 256, 256, 1, (0.0, 1.0, 3.0), (0.0, 0.5, 0.0),
              description="Single-sphere scene, 256x256, 1 spp"),
    2: Config("config2_cornell_8_spheres", 1920, 1080, 4, (0.0, 1.5, 5.5), (0.0, 1.2, 0.0),
              description="8-sphere Cornell-style scene, 1920x1080, 4 spp",
              extra={"sun_azimuth": 1.5707963267948966, "sun_elevation": 0.6}),
    3: Config("config3_10k_spheres", 1920, 1080, 16, (0.0, 6.0, 26.0), (0.0, 1.0, 0.0),
              description="10k random spheres (10 001 single-sphere instances under the TLAS), 1920x1080, 16 spp"),
    4: Config("config4_blob_100k_tris", 3840, 2160, 64, (0.0, 1.6, 3.6), (0.0, 1.05, 0.0),
              description="100 352-triangle displaced sphere + ground, 3840x2160, 64 spp", extra={"nu": 224, "nv": 224}),
    5: Config("config5_terrain_1m_tris", 3840, 2160, 256, (0.0, 7.0, 17.0), (0.0, 0.5, 0.0),
              description="1 002 528-triangle height-field terrain, 3840x2160, 256 spp", extra={"n": 708}),
}


def build_config1(b):
    sid = b.add_sphere(sphere((0.0, 0.5, 0.0), 0.5, (0.8, 0.3, 0.3)))
    b.build_sphere_instance([sid])
    b.rebuild_tlas()


def build_config2(b):
    """Cornell-style box made of 8 spheres, each its own instance (as BuildDefaultScene does):
    floor, back, red left and green right walls (r = 1000, box 4 wide x 4 deep around the
    origin, open top and open front), plus a Lambert, a mirror, a glass and a small Lambert
    sphere.  r = 1000 walls are effectively infinite, so the sun only reaches the floor along
    the open directions: the config's sun comes in through the open front (azimuth pi/2)."""
    white, red, green = (0.75, 0.75, 0.75), (0.75, 0.25, 0.25), (0.25, 0.75, 0.25)
    R = 1000.0
    specs = [
        ((0.0, -R, 0.0), R, white, T.SHADING_LAMBERT, 1.0),          # floor   y = 0
        ((0.0, 1.5, -2.0 - R), R, white, T.SHADING_LAMBERT, 1.0),    # back    z = -2
        ((-2.0 - R, 1.5, 0.0), R, red, T.SHADING_LAMBERT, 1.0),      # left    x = -2
        ((2.0 + R, 1.5, 0.0), R, green, T.SHADING_LAMBERT, 1.0),     # right   x = +2
        ((-1.0, 0.6, -0.6), 0.6, (0.8, 0.8, 0.8), T.SHADING_LAMBERT, 1.0),
        ((1.0, 0.6, -0.3), 0.6, (1.0, 1.0, 1.0), T.SHADING_MIRROR, 1.0),
        ((0.0, 0.5, 0.9), 0.5, (1.0, 1.0, 1.0), T.SHADING_GLASS, 1.5),
        ((-0.35, 0.3, -1.35), 0.3, (0.3, 0.4, 0.9), T.SHADING_LAMBERT, 1.0),
    ]
    ids = [b.add_sphere(sphere(c, r, a, sh, ior)) for c, r, a, sh, ior in specs]
    for i in ids:
        b.build_sphere_instance([i])
    b.rebuild_tlas()


def build_random_spheres(b, count, seed=0x9E3779B9, extent=20.0):
    """ground + `count` random spheres, one instance each (config 3 at count=10000)."""
    rng = XorShift32(seed)
    ids = [b.add_sphere(sphere((0.0, -1000.0, 0.0), 1000.0, (0.6, 0.6, 0.6)))]
    for i in range(count):
        cx = rng.uniform(-extent, extent)
        cy = rng.uniform(0.05, 3.0)
        cz = rng.uniform(-extent, extent)
        r = rng.uniform(0.05, 0.25)
        k = i % 10
        if k == 0:
            ids.append(b.add_sphere(sphere((cx, cy, cz), r, (0.95, 0.95, 0.95), T.SHADING_MIRROR, 1.0)))
        elif k == 1:
            ids.append(b.add_sphere(sphere((cx, cy, cz), r, (1.0, 1.0, 1.0), T.SHADING_GLASS, 1.5)))
        else:
            kd = (rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.9))
            ids.append(b.add_sphere(sphere((cx, cy, cz), r, kd)))
    for i in ids:
        b.build_sphere_instance([i])
    b.rebuild_tlas()


def build_config3(b):
    build_random_spheres(b, 10000)


def grid_mesh(px, py, pz, u, v, kd=(0.8, 0.8, 0.8)):
    """(nv+1) x (nu+1) vertex grid -> 2 triangles per quad, texcoords = (u, v)."""
    nv1, nu1 = px.shape
    pos = np.stack([px, py, pz], axis=-1).astype(np.float32).reshape(-1, 3)
    tex = np.stack([u, v], axis=-1).astype(np.float32).reshape(-1, 2)
    j, i = np.meshgrid(np.arange(nv1 - 1), np.arange(nu1 - 1), indexing="ij")
    a = (j * nu1 + i).reshape(-1)
    b_ = a + 1
    c = a + nu1
    d = c + 1
    tris = np.empty((a.size * 2, 3), dtype=np.int32)
    tris[0::2] = np.stack([a, c, b_], axis=-1)
    tris[1::2] = np.stack([b_, c, d], axis=-1)
    return MeshData(pos, tris, tex, tris.copy(), [material(kd=kd)])


def blob_mesh(nu=224, nv=224, center=(0.0, 1.1, 0.0), radius=1.0, amp=0.15):
    theta = np.linspace(0.0, math.pi, nv + 1)[:, None] * np.ones((1, nu + 1))
    phi = np.ones((nv + 1, 1)) * np.linspace(0.0, 2.0 * math.pi, nu + 1)[None, :]
    r = radius + amp * np.sin(5.0 * theta) * np.sin(7.0 * phi)
    x = center[0] + r * np.sin(theta) * np.cos(phi)
    y = center[1] + r * np.cos(theta)
    z = center[2] + r * np.sin(theta) * np.sin(phi)
    return grid_mesh(x, y, z, phi / (2.0 * math.pi), theta / math.pi)


def terrain_mesh(n=708, half=12.0):
    s = np.linspace(-half, half, n + 1)
    zz, xx = np.meshgrid(s, s, indexing="ij")
    h = np.zeros_like(xx)
    amp, freq = 1.2, 0.35
    for o in range(5):
        h += amp * np.sin(freq * xx + 1.3 * o) * np.sin(freq * zz * 1.1 + 0.7 * o)
        amp *= 0.5
        freq *= 2.0
    return grid_mesh(xx, h, zz, (xx + half) / (2 * half), (zz + half) / (2 * half), kd=(0.7, 0.75, 0.65))


def build_mesh_scene(b, mesh, ground_y=0.0):
    gid = b.add_sphere(sphere((0.0, ground_y - 1000.0, 0.0), 1000.0, (0.6, 0.6, 0.6)))
    b.build_sphere_instance([gid])
    b.load_mesh_instance(mesh)          # LoadObjInstance path rebuilds the TLAS itself


def build_config4(b, nu=224, nv=224):
    build_mesh_scene(b, blob_mesh(nu, nv), ground_y=0.0)


def build_config5(b, n=708):
    build_mesh_scene(b, terrain_mesh(n), ground_y=-3.0)


def build_textured_test_scene(b):
    """Small scene that exercises every branch the benchmark scenes do not: textured sphere
    (atan2/acos UV path), textured + alpha-cut-out two-sided triangles, a scaled/translated
    instance transform, a multi-sphere BLAS.  Used by parity tests only."""
    yy, xx = np.mgrid[0:32, 0:32]
    chk = ((xx // 4 + yy // 4) & 1).astype(np.uint8)
    tex = np.zeros((32, 32, 4), np.uint8)
    tex[..., 0] = 40 + 200 * chk
    tex[..., 1] = 220 - 150 * chk
    tex[..., 2] = 90
    tex[..., 3] = 255
    t0 = b.add_texture(tex)
    ground = b.add_sphere(sphere((0.0, -500.0, 0.0), 500.0, (1.0, 1.0, 1.0), mat=material(kd=(1, 1, 1), diffuse_tex=t0)))
    s_tex = b.add_sphere(sphere((-1.2, 0.6, 0.0), 0.6, (1.0, 1.0, 1.0), mat=material(kd=(1, 1, 1), diffuse_tex=t0)))
    s_mir = b.add_sphere(sphere((1.3, 0.5, 0.3), 0.5, (0.9, 0.9, 0.9), T.SHADING_MIRROR))
    s_gls = b.add_sphere(sphere((0.1, 0.45, 1.1), 0.45, (1.0, 1.0, 1.0), T.SHADING_GLASS, 1.45))
    cluster = [b.add_sphere(sphere((-0.6 + 0.3 * i, 0.15 + 0.05 * (i % 3), -1.4 + 0.1 * (i % 2)), 0.14, (0.3 + 0.1 * i, 0.8 - 0.1 * i, 0.5)))
               for i in range(4)]
    for sid in (ground, s_tex, s_mir, s_gls):
        b.build_sphere_instance([sid])
    b.build_sphere_instance(cluster)                      # 4 spheres, one leaf: F4 quirk cannot fire (<= 4)
    xf = T.Affine3x4()
    sc = 0.5
    xf.m00 = xf.m11 = xf.m22 = sc
    xf.m03, xf.m13, xf.m23 = 0.4, 1.6, -0.5
    moved = b.add_sphere(sphere((0.0, 0.0, 0.0), 0.5, (0.2, 0.4, 0.9)))
    b.build_sphere_instance([moved], xf)
    # alpha-masked, two-sided, textured quad made of 8x8 quads
    n = 8
    s = np.linspace(-1.0, 1.0, n + 1)
    yq, xq = np.meshgrid(s * 0.8 + 1.0, s * 0.9, indexing="ij")
    zq = -1.0 + 0.15 * np.sin(3.0 * xq)
    uq = (xq + 0.9) / 1.8
    vq = (yq - 0.2) / 1.6
    quad = grid_mesh(xq, yq, zq, uq, vq)
    alpha = np.zeros((16, 16, 4), np.uint8)
    ay, ax = np.mgrid[0:16, 0:16]
    g = (((ax - 7.5) ** 2 + (ay - 7.5) ** 2) < 36).astype(np.uint8) * 255
    soft = np.clip(255 - 8 * ((ax - 7.5) ** 2 + (ay - 7.5) ** 2), 0, 255).astype(np.uint8)
    alpha[..., 0] = np.maximum(g // 2, soft)
    alpha[..., 1] = alpha[..., 0]
    alpha[..., 2] = alpha[..., 0]
    alpha[..., 3] = 255
    diffuse = tex[..., [2, 1, 0, 3]]                       # TextureSrc is BGRA
    m = material(kd=(0.9, 0.9, 0.9), diffuse_tex=0, alpha_tex=1, two_sided=1, alpha_cutoff=0.5)
    quad2 = MeshData(quad.positions, quad.triangles, quad.texcoords, quad.tri_uvs, [m], None, [diffuse, alpha])
    b.load_mesh_instance(quad2)


def rotation_affine(axis, degrees, scale=1.0, translate=(0.0, 0.0, 0.0)):
    """Uniform scale * rotation about a coordinate axis + translation: the transforms Scene.InvertRigidOrUniform accepts."""
    a = math.radians(degrees)
    c, s_ = np.float32(math.cos(a)), np.float32(math.sin(a))
    r = np.eye(3, dtype=np.float32)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    r[i, i] = c; r[j, j] = c; r[i, j] = -s_; r[j, i] = s_
    r = (r * np.float32(scale)).astype(np.float32)
    m = T.Affine3x4()
    (m.m00, m.m01, m.m02), (m.m10, m.m11, m.m12), (m.m20, m.m21, m.m22) = [tuple(float(v) for v in row) for row in r]
    m.m03, m.m13, m.m23 = [float(v) for v in translate]
    return m


def build_rotated_instances_scene(b):
    """Rotated + uniformly scaled instances (TransformRay / TransformVector with a full 3x3, tWorld = tObj / scale, world
    normal = normalize(objectToWorld * n)): a triangle grid, a multi-sphere BLAS and a single sphere, over a ground sphere."""
    # all spheres first, instances afterwards, as Scene.BuildDefaultScene does: BuildBLAS_Spheres looks spheres up through the
    # prim-index list by position (Scene.cs:386-390), which only equals the sphere id while no leaf has appended to that list
    g = b.add_sphere(sphere((0.0, -300.0, 0.0), 300.0, (0.8, 0.8, 0.75)))
    ids = [b.add_sphere(sphere((0.5 * i - 1.0, 0.2 * (i % 2), 0.3 * (i % 3)), 0.22, (0.9 - 0.1 * i, 0.3 + 0.1 * i, 0.4),
                               T.SHADING_MIRROR if i == 2 else 0)) for i in range(6)]
    gl = b.add_sphere(sphere((0.0, 0.0, 0.0), 0.5, (1.0, 1.0, 1.0), T.SHADING_GLASS, 1.5))
    b.build_sphere_instance([g])
    b.build_sphere_instance(ids, rotation_affine("z", -25.0, 0.8, (1.4, 0.9, 0.4)))
    b.build_sphere_instance([gl], rotation_affine("x", 60.0, 1.2, (0.1, 0.65, 1.4)))
    s = np.linspace(-1.0, 1.0, 7)
    yq, xq = np.meshgrid(s, s, indexing="ij")
    quad = grid_mesh(xq, yq, 0.2 * np.sin(2.0 * xq) * np.cos(1.5 * yq), (xq + 1) / 2, (yq + 1) / 2, kd=(0.3, 0.6, 0.9))
    b.load_mesh_instance(quad, rotation_affine("y", 35.0, 1.4, (-1.2, 1.5, -0.6)))


def build(cfg_id, b, **kw):
    {1: build_config1, 2: build_config2, 3: build_config3, 4: build_config4, 5: build_config5}[cfg_id](b, **kw)
    return CONFIGS[cfg_id]


def frame_params(cfg, make_camera, bake, sun_dir, width=None, height=None, spp=None, frame=0,
                 reuse=False, rng_lock_noise=0, prev_cam=None):
    """Common frame parameters of SURVEY.md 8d.  `make_camera(origin, lookat, up, vfov, aspect)`,
    `bake(cam, w, h)` and `sun_dir(az, el)` come from the host under test (product or oracle)."""
    w = width or cfg.width
    h = height or cfg.height
    cam = make_camera(cfg.cam_origin, cfg.cam_lookat, (0.0, 1.0, 0.0), cfg.vfov, float(np.float32(w) / np.float32(h)))
    bake(cam, w, h)
    p = T.FrameParams()
    p.width, p.height, p.frame = w, h, frame
    p.cam = cam
    p.prevCam = prev_cam if prev_cam is not None else cam
    p.dirLightDir = T.f3(*sun_dir(cfg.extra.get("sun_azimuth", 0.0), cfg.extra.get("sun_elevation", 0.9)))
    p.dirLightRadiance = T.f3(10.0, 10.0, 10.0)
    p.skyTintTop = T.f3(0.5, 0.7, 1.0)
    p.skyTintBottom = T.f3(1.0, 1.0, 1.0)
    p.debugCamSeq = 0
    p.enableTemporalReuse = 1 if reuse else 0
    p.enableSpatialReuse = 1 if reuse else 0
    p.rngLockNoise = rng_lock_noise
    p.spp = spp if spp is not None else cfg.spp
    p.maxDepth = cfg.max_depth
    return p
