#!/usr/bin/env python3
"""CPU model of the treelet structure of BLAS walks (round 3, before the LDS-staged walker was built).

For secondary rays of config 4 / 5 (shadow rays towards the sun and cosine-distributed bounce rays leaving the primary hit
points of random pixels) it walks the mesh BLAS in walk order, as the device does, and reports per ray
  * node visits and triangle tests,
  * how they split between the TOP of the tree (nodes whose subtree exceeds a treelet) and bottom TREELETS (maximal subtrees
    of at most S nodes),
  * how many distinct treelets a ray enters, and which share of its work lies in the first one / in the treelet that holds
    the triangle the ray starts from.
    python tools/treelet_model.py [config] [S] [nrays]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import engine, scenes

cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 511
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 20000

s = engine.Scene()
scenes.build(cfg_id, s)
cfg = scenes.CONFIGS[cfg_id]
A = s.arrays()
inst = A["instances"]
mi = [i for i in range(len(inst)) if inst[i]["type"] == 2][0]
root, cnt = int(inst[mi]["blasRoot"]), int(inst[mi]["blasNodeCount"])
N = A["blasNodes"][root:root + cnt]
blo = np.stack([N["boundsMin"][f] for f in "XYZ"], 1).astype(np.float32)
bhi = np.stack([N["boundsMax"][f] for f in "XYZ"], 1).astype(np.float32)
bskip = N["skipIndex"].astype(np.int64); bleft = N["left"].astype(np.int64); bcount = N["count"].astype(np.int64); bfirst = N["first"].astype(np.int64)
bskip = np.where(bskip < 0, -1, bskip - root); bleft = np.where(bcount > 0, -1, bleft - root)
# walk order
perm = -np.ones(cnt, np.int64); order = []
st = [0]
while st:
    i = st.pop()
    if i < 0 or perm[i] >= 0: continue
    perm[i] = len(order); order.append(i)
    st.append(bskip[i])
    if bcount[i] == 0: st.append(bleft[i])
order = np.array(order); n = len(order)
lo = blo[order]; hi = bhi[order]; count = bcount[order]; first = bfirst[order]
skip = np.where(bskip[order] < 0, n, perm[np.maximum(bskip[order], 0)])
sub = skip - np.arange(n)                              # subtree size (nodes)
# triangles per leaf slot
tpi = A["triPrimIdx"]; tris = A["meshTris"]; pos = np.stack([A["meshPositions"][f] for f in "XYZ"], 1).astype(np.float32)
tv = np.stack([tris[f] for f in ("i0", "i1", "i2")], 1)[tpi]       # per leaf slot
V0, V1, V2 = pos[tv[:, 0]], pos[tv[:, 1]], pos[tv[:, 2]]
# treelets: maximal subtrees with <= S nodes
parent = -np.ones(n, np.int64)
for i in range(n):
    if count[i] == 0:
        c = i + 1
        while c < skip[i]: parent[c] = i; c = skip[c]
is_top = sub > S
troot = np.flatnonzero(~is_top & ((parent < 0) | is_top[np.maximum(parent, 0)]))
tid = -np.ones(n, np.int64)
for k, r in enumerate(troot): tid[r:skip[r]] = k
leaf_nodes = np.flatnonzero(count > 0)
is_troot = np.zeros(n, bool); is_troot[troot] = True
slot_tid = -np.ones(len(tpi), np.int64)
for i in leaf_nodes: slot_tid[first[i]:first[i] + count[i]] = tid[i]
tl_tris = np.array([count[r:skip[r]].sum() for r in troot])
print("config %d: %d nodes, %d leaves, %.2f tris/leaf; S=%d -> %d treelets (nodes %d..%d, tris %d..%d), top nodes %d; bytes/treelet max %d"
      % (cfg_id, n, len(leaf_nodes), count[leaf_nodes].mean(), S, len(troot), sub[troot].min(), sub[troot].max(), tl_tris.min(), tl_tris.max(), int(is_top.sum()),
         int((sub[troot] * 32 + tl_tris * 48).max())))


def walk(o, d, anyhit, tmax=1e30):
    """vectorised skip-link walk; returns t, prim slot, and a per-ray list of visited node ids (as arrays per step)"""
    R = len(o)
    inv = (1.0 / np.where(d != 0, d, np.float32(1e-8))).astype(np.float32)
    cur = np.zeros(R, np.int64); best = np.full(R, tmax, np.float32); prim = -np.ones(R, np.int64)
    done = np.zeros(R, bool); log_nodes = []; tri_tests = np.zeros(R, np.int64); roothits = np.zeros(R, np.int64)
    while True:
        act = ~done & (cur < n)
        if not act.any(): break
        idx = np.flatnonzero(act); c = cur[idx]
        log = -np.ones(R, np.int64); log[idx] = c; log_nodes.append(log)
        t1 = (lo[c] - o[idx]) * inv[idx]; t2 = (hi[c] - o[idx]) * inv[idx]
        tmn = np.minimum(t1, t2).max(1); tmx = np.maximum(t1, t2).min(1)
        hit = (tmx >= np.maximum(tmn, np.float32(0.001))) & (tmn <= best[idx])
        isleaf = count[c] > 0
        roothits[idx[hit & is_troot[c]]] += 1
        nxt = np.where(hit & ~isleaf, c + 1, skip[c])
        cur[idx] = nxt
        lf = idx[hit & isleaf]; lc = c[hit & isleaf]
        for j in range(4):
            m = count[lc] > j
            if not m.any(): break
            r = lf[m]; sl = first[lc[m]] + j
            if anyhit: keep = ~done[r]; r = r[keep]; sl = sl[keep]
            tri_tests[r] += 1
            e1 = V1[sl] - V0[sl]; e2 = V2[sl] - V0[sl]
            p = np.cross(d[r], e2); det = (e1 * p).sum(1)
            ok = np.abs(det) >= 1e-8
            idet = 1.0 / np.where(ok, det, 1)
            tv_ = o[r] - V0[sl]; u = (tv_ * p).sum(1) * idet
            q = np.cross(tv_, e1); v = (d[r] * q).sum(1) * idet
            t = (e2 * q).sum(1) * idet
            ok &= (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0.001) & (t < best[r])
            best[r[ok]] = t[ok].astype(np.float32); prim[r[ok]] = sl[ok]
            if anyhit: done[r[ok]] = True
    walk.roothits = roothits
    return best, prim, np.stack(log_nodes, 1), tri_tests


rng = np.random.default_rng(5)
eye = np.array(cfg.cam_origin, np.float32); look = np.array(cfg.cam_lookat, np.float32)
fw = (look - eye) / np.linalg.norm(look - eye); right = np.cross(fw, [0, 1, 0]); right /= np.linalg.norm(right); up = np.cross(right, fw)
th = math.tan(math.radians(60.0) / 2)
D = fw[None] + (rng.uniform(-1, 1, NR) * th * 16 / 9)[:, None] * right[None] + (rng.uniform(-1, 1, NR) * th)[:, None] * up[None]
D = (D / np.linalg.norm(D, axis=1)[:, None]).astype(np.float32)
O = np.repeat(eye[None], NR, 0)
t, prim, logn, _ = walk(O, D, False)
hitm = prim >= 0
print("primary rays: %d of %d hit the mesh; node visits per ray %.1f" % (hitm.sum(), NR, (logn >= 0).sum() / NR))
O = O[hitm]; D = D[hitm]; t = t[hitm]; prim = prim[hitm]
P = O + D * t[:, None]
nrm = np.cross(V1[prim] - V0[prim], V2[prim] - V0[prim]); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
nrm = np.where(((nrm * D).sum(1) > 0)[:, None], -nrm, nrm).astype(np.float32)
Po = (P + nrm * np.float32(0.0025)).astype(np.float32)
sun = np.array([math.cos(0.9) * math.cos(0.0), math.sin(0.9), math.cos(0.9) * math.sin(0.0)], np.float32)
sun = np.array([0.6216, 0.7833, 0.0], np.float32)


def cos_dirs(nrm):
    r1 = rng.uniform(0, 1, len(nrm)); r2 = rng.uniform(0, 1, len(nrm))
    phi = 2 * math.pi * r1; r = np.sqrt(r2)
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, [[0, 1, 0]], [[1, 0, 0]])
    tt = np.cross(nrm, a); tt /= np.linalg.norm(tt, axis=1)[:, None]; bb = np.cross(nrm, tt)
    return (tt * (r * np.cos(phi))[:, None] + bb * (r * np.sin(phi))[:, None] + nrm * np.sqrt(1 - r2)[:, None]).astype(np.float32)


def report(name, logn, tri_tests, origin_tid):
    R = logn.shape[0]
    rh = walk.roothits
    print("%s: PORTAL HITS per ray mean %.2f; P(>=1) %.3f P(>=2|>=1) %.3f P(>=3|>=2) %.3f P(>=4|>=3) %.3f" % (name, rh.mean(), (rh >= 1).mean(), (rh >= 2).sum() / max((rh >= 1).sum(), 1), (rh >= 3).sum() / max((rh >= 2).sum(), 1), (rh >= 4).sum() / max((rh >= 3).sum(), 1)))
    vis = logn >= 0
    nv = vis.sum(1)
    tids = np.where(vis, tid[np.maximum(logn, 0)], -2)        # -1 top, -2 none
    top = (tids == -1).sum(1); bot = (tids >= 0).sum(1)
    # distinct treelets entered, in order
    ent = np.zeros(R, np.int64); first_share = np.zeros(R); orig_share = np.zeros(R); hops = np.zeros(R, np.int64)
    for r in range(R):
        x = tids[r][tids[r] >= 0]
        if len(x) == 0: continue
        ch = np.flatnonzero(np.diff(x) != 0)
        ent[r] = len(ch) + 1
        runs = np.diff(np.concatenate([[-1], ch, [len(x) - 1]]))
        hops[r] = (runs > 1).sum()                       # treelets whose root box was hit (the walk went on inside)
        first_share[r] = (x == x[0]).sum() / len(x)
        orig_share[r] = (x == origin_tid[r]).sum() / len(x)
    print("%s: %d rays | node visits %.1f (top %.1f, treelets %.1f) | tri tests %.1f | treelets entered: mean %.2f, p50 %d, p90 %d, max %d | rays entering none %.1f %%"
          % (name, R, nv.mean(), top.mean(), bot.mean(), tri_tests.mean(), ent.mean(), np.percentile(ent, 50), np.percentile(ent, 90), ent.max(), 100.0 * (ent == 0).mean()))
    w = bot.sum()
    print("    share of treelet node visits in the FIRST treelet entered: %.1f %%; in the ORIGIN treelet (holds the triangle the ray leaves): %.1f %%"
          % (100.0 * (first_share * bot).sum() / w, 100.0 * (orig_share * bot).sum() / w))
    print("    treelets ENTERED PAST THE ROOT (a hop of the queued design): mean %.2f, p90 %d, max %d; rays by hops: %s"
          % (hops.mean(), np.percentile(hops, 90), hops.max(), " ".join("%d:%.1f%%" % (k, 100.0 * v / R) for k, v in enumerate(np.bincount(hops)) if v)))
    hist = np.bincount(ent, minlength=8)
    print("    rays by treelets entered:", " ".join("%d:%.1f%%" % (k, 100.0 * hist[k] / R) for k in range(len(hist)) if hist[k]))


# primary rays that miss the mesh and hit the ground sphere (centre (0, gy - 1000, 0), radius 1000): their secondary rays cross the mesh's boxes too
gy = 0.0 if cfg_id == 4 else -3.0
Og = np.repeat(eye[None], NR, 0)[~hitm]; Dg = (fw[None] + 0 * Og)  # placeholder, replaced below
rng2 = np.random.default_rng(5)
Dall = fw[None] + (rng2.uniform(-1, 1, NR) * th * 16 / 9)[:, None] * right[None] + (rng2.uniform(-1, 1, NR) * th)[:, None] * up[None]
Dall = (Dall / np.linalg.norm(Dall, axis=1)[:, None]).astype(np.float32)
Dg = Dall[~hitm]
c = np.array([0.0, gy - 1000.0, 0.0], np.float32)
oc = Og - c; b = (oc * Dg).sum(1); cc = (oc * oc).sum(1) - 1000.0 ** 2; disc = b * b - cc
okg = (disc > 0) & ((-b - np.sqrt(np.maximum(disc, 0))) > 0.001)
tg = (-b - np.sqrt(np.maximum(disc, 0)))[okg]
Pg = Og[okg] + Dg[okg] * tg[:, None]; ng = (Pg - c) / 1000.0
Pgo = (Pg + ng * np.float32(0.0025)).astype(np.float32); ng = ng.astype(np.float32)
print("primary rays: %d hit the ground, %d the sky" % (okg.sum(), (~okg).sum()))
otid = slot_tid[prim]
ts, ps, logs, tts = walk(Po, np.repeat(sun[None], len(Po), 0), True, 1e29)
print("shadow rays occluded: %.1f %%" % (100.0 * (ps >= 0).mean()))
report("shadow", logs, tts, otid)
tb, pb, logb, ttb = walk(Po, cos_dirs(nrm), False)
print("bounce rays that hit the mesh again: %.1f %%" % (100.0 * (pb >= 0).mean()))
report("bounce", logb, ttb, otid)

ogt = -np.ones(len(Pgo), np.int64)
ts, ps, logs, tts = walk(Pgo, np.repeat(sun[None], len(Pgo), 0), True, 1e29)
report("shadow from the ground", logs, tts, ogt)
tb, pb, logb, ttb = walk(Pgo, cos_dirs(ng), False)
report("bounce from the ground", logb, ttb, ogt)
