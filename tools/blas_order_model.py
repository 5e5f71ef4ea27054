#!/usr/bin/env python3
"""CPU model: box tests of an ANY-HIT walk through config 4's mesh BLAS (100 352 triangles, 32 768 leaves) on the reference's median-split
tree and on a binned-SAH tree over the SAME leaves and leaf boxes (any-hit walks do not depend on the hierarchy: DESIGN.md 8).
Result (round 2): no difference -- 117.4 against 117.4 box tests per shadow ray, the same leaves entered: on a uniformly
tessellated mesh the median split is as good as it gets.   python tools/blas_order_model.py"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import engine, scenes
sys.setrecursionlimit(100000)
s = engine.Scene(); scenes.build_config4(s, 224, 224)
A = s.arrays()
inst = A["instances"]
mi = [i for i in range(len(inst)) if inst[i]["type"] == 2][0]
root, cnt = int(inst[mi]["blasRoot"]), int(inst[mi]["blasNodeCount"])
N = A["blasNodes"][root:root + cnt]
lo = np.stack([N["boundsMin"][f] for f in "XYZ"], 1).astype(np.float64); hi = np.stack([N["boundsMax"][f] for f in "XYZ"], 1).astype(np.float64)
skip = N["skipIndex"].astype(np.int64); left = N["left"].astype(np.int64); count = N["count"].astype(np.int64)
skip = np.where(skip < 0, root + cnt, skip) - root; left = left - root
leaves = np.flatnonzero(count > 0)
print("nodes", cnt, "leaves", len(leaves), "tris/leaf", count[leaves].mean())
# reference walk (any-hit without primitives: count box tests, leaf boxes hit)
def walk_ref(o, d):
    inv = 1.0 / np.where(d != 0, d, 1e-8)
    cur, visits, lh = 0, 0, 0
    while cur < cnt:
        visits += 1
        t1 = (lo[cur] - o) * inv; t2 = (hi[cur] - o) * inv
        tmn = np.minimum(t1, t2).max(); tmx = np.maximum(t1, t2).min()
        if tmx >= max(tmn, 0.001):
            if count[cur] > 0: lh += 1; cur = skip[cur]
            else: cur = left[cur]
        else: cur = skip[cur]
    return visits, lh
# SAH tree over the leaf boxes
llo, lhi = lo[leaves], hi[leaves]; cen = 0.5 * (llo + lhi)
def area(l, h):
    e = np.maximum(h - l, 0); return 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])
blo, bhi, bskip, bleaf = [], [], [], []
def sah(it, nb=16):
    best, clo, chi = None, cen[it].min(0), cen[it].max(0)
    for ax in range(3):
        if chi[ax] <= clo[ax]: continue
        b = np.minimum(((cen[it, ax] - clo[ax]) / (chi[ax] - clo[ax]) * nb).astype(int), nb - 1)
        for k in range(1, nb):
            L, R = it[b < k], it[b >= k]
            if len(L) == 0 or len(R) == 0: continue
            c = area(llo[L].min(0), lhi[L].max(0)) * len(L) + area(llo[R].min(0), lhi[R].max(0)) * len(R)
            if best is None or c < best[0]: best = (c, L, R)
    return (it[:len(it) // 2], it[len(it) // 2:]) if best is None else (best[1], best[2])
def rec(it):
    i = len(blo); blo.append(llo[it].min(0)); bhi.append(lhi[it].max(0)); bskip.append(-1); bleaf.append(len(it) == 1)
    if len(it) > 1:
        a, b = sah(it); rec(a); rec(b)
    bskip[i] = len(blo)
t0 = time.time(); rec(np.arange(len(leaves))); print("sah built", time.time() - t0)
blo, bhi, bskip, bleaf = np.array(blo), np.array(bhi), np.array(bskip), np.array(bleaf)
def walk_sah(o, d):
    inv = 1.0 / np.where(d != 0, d, 1e-8)
    cur, visits, lh, n = 0, 0, 0, len(bskip)
    while cur < n:
        visits += 1
        t1 = (blo[cur] - o) * inv; t2 = (bhi[cur] - o) * inv
        tmn = np.minimum(t1, t2).max(); tmx = np.maximum(t1, t2).min()
        if tmx >= max(tmn, 0.001):
            if bleaf[cur]: lh += 1; cur = bskip[cur]
            else: cur += 1
        else: cur = bskip[cur]
    return visits, lh
rng = np.random.default_rng(2)
cfg = scenes.CONFIGS[4]
eye = np.array(cfg.cam_origin, float); look = np.array(cfg.cam_lookat, float)
fw = (look - eye) / np.linalg.norm(look - eye); right = np.cross(fw, [0, 1, 0]); right /= np.linalg.norm(right); up = np.cross(right, fw)
th = np.tan(np.radians(60.0) / 2); n = 300
D = fw[None] + (rng.uniform(-1, 1, n) * th * 16 / 9)[:, None] * right[None] + (rng.uniform(-1, 1, n) * th)[:, None] * up[None]
D /= np.linalg.norm(D, axis=1)[:, None]
# object space = world here? use instance transform identity check
print("o2w", [float(inst[mi]["objectToWorld"]["m%d%d" % (r, c)]) for r in range(3) for c in range(4)])
sun = np.array([0.3, 0.8, 0.5]); sun /= np.linalg.norm(sun)
pos = np.stack([A["meshPositions"][f] for f in "XYZ"], 1).astype(np.float64)
P = pos[rng.integers(0, len(pos), n)]
nrm = P - np.array([0.0, 1.1, 0.0]); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
O = P + 1e-3 * nrm
for name, rays in (("camera rays", [(eye, d) for d in D]), ("shadow rays from the surface", [(o, sun) for o in O]), ("random rays from the surface", [(o, (lambda r: r / np.linalg.norm(r))(rng.normal(size=3))) for o in O])):
    a = np.array([walk_ref(o, d) for o, d in rays]); b = np.array([walk_sah(o, d) for o, d in rays])
    print("%-30s reference tree: %.1f box tests, %.2f leaves entered | SAH over the same leaves: %.1f box tests, %.2f leaves" % (name, a[:, 0].mean(), a[:, 1].mean(), b[:, 0].mean(), b[:, 1].mean()))
