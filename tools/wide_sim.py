"""Design study (CPU, numpy): leaf-order-preserving wide BVH over the reference's binary tree.

Counts, for a sample of real rays of a config, what a walk fetches under
  (a) the reference's binary skip-pointer tree (one 32-byte node per visit), and
  (b) a k-wide tree obtained by collapsing the same binary tree (children kept in the reference's DFS order,
      child boxes quantised to 8 bits inside the parent box, rounded outward), with the exact 32-byte box of a
      reference leaf fetched only when its quantised box is hit.
Both use the final hit distance as tMax from the start (perfect pruning), so the numbers are lower bounds with the
same bias on both sides.  Usage: python tools/wide_sim.py [config] [width] [n_rays]
"""
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilgpu_raytracing_amd import engine, scenes
from oracle import orc


def slab(o, inv, lo, hi, tmin, tmax):
    t0 = (lo - o) * inv
    t1 = (hi - o) * inv
    near = np.minimum(t0, t1).max()
    far = np.maximum(t0, t1).min()
    return far >= max(near, tmin) and near <= tmax


def main():
    cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    s = engine.Scene()
    cfg = scenes.build(cfg_id, s)
    so = orc.OrcScene()
    scenes.build(cfg_id, so)
    a = s.arrays()
    if cfg_id >= 4:
        inst = a["instances"][-1] if a["instances"]["type"][-1] == 2 else a["instances"][0]
        nodes = a["blasNodes"][inst["blasRoot"]:inst["blasRoot"] + inst["blasNodeCount"]]
        base = int(inst["blasRoot"])
    else:
        nodes = a["tlasNodes"]
        base = 0
    n = len(nodes)
    bmin = np.stack([nodes["boundsMin"][c] for c in "XYZ"], 1).astype(np.float64)
    bmax = np.stack([nodes["boundsMax"][c] for c in "XYZ"], 1).astype(np.float64)
    left, right, cnt, skip = nodes["left"] - base, nodes["right"] - base, nodes["count"], nodes["skipIndex"]
    skip = np.where(skip >= 0, skip - base, -1)
    print("config", cfg_id, "binary nodes", n, "leaves", int((cnt > 0).sum()))

    # ---- rays: camera rays, then cosine bounces and sun shadow rays from their hits
    rng = np.random.RandomState(1)
    w, h = 160, 90
    cam = engine.camera_look_at(cfg.cam_origin, cfg.cam_lookat, (0, 1, 0), cfg.vfov, w / h)
    ll = np.array([cam.lowerLeft.X, cam.lowerLeft.Y, cam.lowerLeft.Z]); hz = np.array([cam.horizontal.X, cam.horizontal.Y, cam.horizontal.Z])
    vt = np.array([cam.vertical.X, cam.vertical.Y, cam.vertical.Z]); org = np.array([cam.origin.X, cam.origin.Y, cam.origin.Z])
    px = rng.randint(0, w, n_rays); py = rng.randint(0, h, n_rays)
    d = ll + ((px + 0.5) / w)[:, None] * hz + ((py + 0.5) / h)[:, None] * vt - org
    d /= np.linalg.norm(d, axis=1)[:, None]
    o = np.repeat(org[None], n_rays, 0)
    r1 = orc.trace_rays(so.desc(), o, d)
    hit = r1["hit"] != 0
    p = o[hit] + d[hit] * r1["t"][hit][:, None]
    nn = r1["normal"][hit].astype(np.float64)
    u1, u2 = rng.rand(len(p)), rng.rand(len(p))
    up = np.where(np.abs(nn[:, 1:2]) < 0.999, np.array([[0, 1, 0.0]]), np.array([[1, 0, 0.0]]))
    tt = np.cross(up, nn); tt /= np.linalg.norm(tt, axis=1)[:, None]
    bb = np.cross(nn, tt)
    ph = 2 * np.pi * u1
    bd = tt * (np.cos(ph) * np.sqrt(u2))[:, None] + bb * (np.sin(ph) * np.sqrt(u2))[:, None] + nn * np.sqrt(1 - u2)[:, None]
    bo = p + nn * 0.0025
    r2 = orc.trace_rays(so.desc(), bo, bd)
    sets = {"primary": (o, d, np.where(hit, r1["t"], 1e30)), "bounce": (bo, bd, np.where(r2["hit"] != 0, r2["t"], 1e30))}

    # ---- k-wide collapse of the binary tree, children in DFS order
    def area(i):
        e = bmax[i] - bmin[i]
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]

    wide = []            # list of (children binary ids)
    wid_of = {}

    def build(root):
        kids = [root]
        while len(kids) < width:
            cand = [(area(k), j) for j, k in enumerate(kids) if cnt[k] == 0]
            if not cand:
                break
            _, j = max(cand)
            k = kids[j]
            kids[j:j + 1] = [int(left[k]), int(right[k])]
        me = len(wide)
        wide.append(kids)
        wid_of[root] = me
        for k in kids:
            if cnt[k] == 0:
                build(k)
        return me

    sys.setrecursionlimit(100000)
    if cnt[0] > 0:
        print("single-leaf tree"); return
    build(0)
    print("wide nodes", len(wide), "avg children", np.mean([len(k) for k in wide]))
    # quantised child boxes
    qlo, qhi = {}, {}
    for root, me in wid_of.items():
        kids = wide[me]
        lo = bmin[kids].min(0); hi = bmax[kids].max(0)
        ext = np.maximum(hi - lo, 1e-30)
        scale = 2.0 ** np.ceil(np.log2(ext / 255.0))
        ql = np.floor((bmin[kids] - lo) / scale) - 1          # one extra quantum of safety on both sides
        qh = np.ceil((bmax[kids] - lo) / scale) + 1
        qlo[me] = lo + ql * scale; qhi[me] = lo + qh * scale

    for name, (O, D, T) in sets.items():
        vis_b = vis_w = child_pass = leaf_cons = leaf_exact = bin_leaf = 0
        maxdepth = 0
        m = len(O)
        for r in range(m):
            oo, dd, tm = O[r].astype(np.float64), D[r].astype(np.float64), float(T[r])
            inv = 1.0 / np.where(dd != 0, dd, 1e-8)
            cur = 0
            while cur >= 0 and cur < n:
                vis_b += 1
                if slab(oo, inv, bmin[cur], bmax[cur], 0.001, tm):
                    if cnt[cur] > 0:
                        bin_leaf += 1; cur = int(skip[cur])
                    else:
                        cur = int(left[cur])
                else:
                    cur = int(skip[cur])
            stack = [(0, 0)]
            while stack:
                maxdepth = max(maxdepth, len(stack))
                me, depth = stack.pop()
                vis_w += 1
                kids = wide[me]
                push = []
                for j, k in enumerate(kids):
                    if slab(oo, inv, qlo[me][j], qhi[me][j], 0.001, tm):
                        child_pass += 1
                        if cnt[k] > 0:
                            leaf_cons += 1
                            if slab(oo, inv, bmin[k], bmax[k], 0.001, tm):
                                leaf_exact += 1
                        else:
                            push.append((wid_of[k], depth + 1))
                stack.extend(reversed(push))
        print("%-8s rays %d | binary: visits/ray %.1f (64 B-lines x2 loads = %.0f lane-loads), leaf passes %.2f | wide%d: nodes/ray %.1f (x%d loads = %.0f), "
              "children hit %.1f, leaf cons %.2f exact %.2f (+%.0f loads), max stack %d"
              % (name, m, vis_b / m, 2 * vis_b / m, bin_leaf / m, width, vis_w / m, (16 + 6 * width + 8 * width // 8 + 15) // 16,
                 vis_w / m * ((16 + 6 * width + 15) // 16), child_pass / m, leaf_cons / m, leaf_exact / m, 2 * leaf_cons / m, maxdepth))


if __name__ == "__main__":
    main()
