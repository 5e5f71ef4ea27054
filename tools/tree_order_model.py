#!/usr/bin/env python3
"""CPU model of the node visits of the fixed-order (skip-link) walk over different SECOND trees of BASELINE config 3 (10 001
one-sphere instances): the reference's median split, the device's 30-bit LBVH, a binned-SAH tree, and the SAH tree kept in eight
child orders, one per ray-direction octant (the near child along the axis that separates the children most comes first).
float64 slab and sphere tests, 400 primary rays of the config's camera, their bounce rays (two generations) and shadow rays.
No GPU.  Result of round 2 in profiles/r02_walk_experiments.txt (14).   usage: python tools/tree_order_model.py [n_primary]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import engine, scenes

s = engine.Scene(); scenes.build_config3(s)
A = s.arrays()
inst = A["instances"]; n = len(inst)
lo = np.stack([inst["worldBoundsMin"][f] for f in "XYZ"], 1).astype(np.float64)
hi = np.stack([inst["worldBoundsMax"][f] for f in "XYZ"], 1).astype(np.float64)
sp, prim = A["spheres"], A["spherePrimIdx"]
sid = np.array([prim[A["blasNodes"][inst[i]["blasRoot"]]["first"]] for i in range(n)])
cen = np.stack([sp["center"][f] for f in "XYZ"], 1).astype(np.float64)[sid]; rad = sp["radius"].astype(np.float64)[sid]
cent = 0.5 * (lo + hi)
sys.setrecursionlimit(100000)


class Tree: pass


def emit(children, items):
    """children(items) -> (first, second) item arrays; leaves hold <= 2 instances.  Arrays in walk order."""
    blo, bhi, skip, first, count, order = [], [], [], [], [], []
    def rec(it):
        i = len(blo)
        blo.append(lo[it].min(0)); bhi.append(hi[it].max(0)); skip.append(-1); first.append(-1); count.append(0)
        if len(it) <= 2:
            first[i] = len(order); count[i] = len(it); order.extend(it.tolist())
        else:
            a, b = children(it); rec(a); rec(b)
        skip[i] = len(blo)
    rec(items)
    t = Tree(); t.blo, t.bhi, t.skip, t.first, t.count, t.order = (np.array(v) for v in (blo, bhi, skip, first, count, order))
    return t


def median_children(it):                      # Scene.cs BuildTLASNodeRecursive: longest axis of the bounds, median of the centroids
    ext = hi[it].max(0) - lo[it].min(0)
    ax = 0 if (ext[0] > ext[1] and ext[0] > ext[2]) else (1 if ext[1] > ext[2] else 2)
    o = it[np.argsort(cent[it, ax], kind="stable")]
    return o[:len(o) // 2], o[len(o) // 2:]


def lbvh_children():                          # hrt_bvh.hip: 10 bits per axis on the cube over the centroids, split at the top differing bit
    cmin = cent.min(0); ext = (cent.max(0) - cmin).max()
    q = np.clip(((cent - cmin) / ext * 1024).astype(np.int64), 0, 1023)
    code = np.zeros(n, np.int64)
    for b in range(10):
        for a in range(3): code |= ((q[:, a] >> b) & 1) << (3 * b + (2 - a))
    def ch(it):
        o = it[np.argsort(code[it], kind="stable")]; c = code[o]
        if c[0] == c[-1]: m = len(o) // 2
        else: m = int(np.searchsorted((c >> (int(c[0] ^ c[-1]).bit_length() - 1)) & 1, 1))
        return o[:m], o[m:]
    return ch


def area(l, h):
    e = np.maximum(h - l, 0); return 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])


def sah_children(it, nb=16):
    best, clo, chi = None, cent[it].min(0), cent[it].max(0)
    for ax in range(3):
        if chi[ax] <= clo[ax]: continue
        b = np.minimum(((cent[it, ax] - clo[ax]) / (chi[ax] - clo[ax]) * nb).astype(int), nb - 1)
        for k in range(1, nb):
            L, R = it[b < k], it[b >= k]
            if len(L) == 0 or len(R) == 0: continue
            cost = area(lo[L].min(0), hi[L].max(0)) * len(L) + area(lo[R].min(0), hi[R].max(0)) * len(R)
            if best is None or cost < best[0]: best = (cost, L, R)
    return (it[:len(it) // 2], it[len(it) // 2:]) if best is None else (best[1], best[2])


def octant_order(fn, sgn):
    def g(it):
        L, R = fn(it)
        cl, cr = 0.5 * (lo[L].min(0) + hi[L].max(0)), 0.5 * (lo[R].min(0) + hi[R].max(0))
        ax = int(np.argmax(np.abs(cl - cr)))
        return (L, R) if ((cl[ax] <= cr[ax]) if sgn[ax] > 0 else (cl[ax] >= cr[ax])) else (R, L)
    return g


def walk(t, o, d, anyhit=False):
    inv = 1.0 / np.where(d != 0, d, 1e-8)
    cur, best, visits, hit, nT = 0, 1e30, 0, -1, len(t.skip)
    while cur < nT:
        visits += 1
        t1, t2 = (t.blo[cur] - o) * inv, (t.bhi[cur] - o) * inv
        tmn, tmx = np.minimum(t1, t2).max(), np.maximum(t1, t2).min()
        if tmx >= max(tmn, 0.001) and tmn <= best:
            if t.count[cur] > 0:
                for k in range(t.count[cur]):
                    ii = t.order[t.first[cur] + k]
                    oc = o - cen[ii]; b = oc @ d; disc = b * b - (oc @ oc - rad[ii] ** 2)
                    if disc > 0:
                        sq = np.sqrt(disc); tt = -b - sq
                        if tt < 0.001: tt = -b + sq
                        if 0.001 < tt < best:
                            best, hit = tt, ii
                            if anyhit: return visits, best, hit
                cur = t.skip[cur]
            else: cur += 1
        else: cur = t.skip[cur]
    return visits, best, hit


def walk_stack(t, o, d, anyhit=False):
    """Ordered traversal with a stack over the same tree: both children of an inner node are tested, the nearer entered first.
    Counts box tests, like walk()."""
    inv = 1.0 / np.where(d != 0, d, 1e-8)
    def box(i, best):
        t1, t2 = (t.blo[i] - o) * inv, (t.bhi[i] - o) * inv
        tmn, tmx = np.minimum(t1, t2).max(), np.maximum(t1, t2).min()
        return tmn if (tmx >= max(tmn, 0.001) and tmn <= best) else None
    best, hit, visits = 1e30, -1, 1
    if box(0, best) is None: return visits, best, hit
    stack = [(0, 0.0)]
    while stack:
        cur, entry = stack.pop()
        if entry > best: continue
        if t.count[cur] > 0:
            for k in range(t.count[cur]):
                ii = t.order[t.first[cur] + k]
                oc = o - cen[ii]; b = oc @ d; disc = b * b - (oc @ oc - rad[ii] ** 2)
                if disc > 0:
                    sq = np.sqrt(disc); tt = -b - sq
                    if tt < 0.001: tt = -b + sq
                    if 0.001 < tt < best:
                        best, hit = tt, ii
                        if anyhit: return visits, best, hit
            continue
        l, r = cur + 1, t.skip[cur + 1]
        visits += 2
        el, er = box(l, best), box(r, best)
        if el is not None and er is not None:
            if el <= er: stack.append((r, er)); stack.append((l, el))
            else: stack.append((l, el)); stack.append((r, er))
        elif el is not None: stack.append((l, el))
        elif er is not None: stack.append((r, er))
    return visits, best, hit


def main():
    n_primary = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    cfg = scenes.CONFIGS[3]
    eye, look = np.array(cfg.cam_origin, float), np.array(cfg.cam_lookat, float)
    fw = (look - eye) / np.linalg.norm(look - eye); right = np.cross(fw, [0, 1, 0]); right /= np.linalg.norm(right); up = np.cross(right, fw)
    rng = np.random.default_rng(1)
    th = np.tan(np.radians(60.0) / 2)
    D = fw[None] + (rng.uniform(-1, 1, n_primary) * th * 16 / 9)[:, None] * right[None] + (rng.uniform(-1, 1, n_primary) * th)[:, None] * up[None]
    D /= np.linalg.norm(D, axis=1)[:, None]
    items = np.arange(n)
    t0 = time.time()
    sah = emit(sah_children, items)
    octs = {o: emit(octant_order(sah_children, [1 if (o >> a) & 1 else -1 for a in range(3)]), items) for o in range(8)}
    pick = lambda d: octs[sum((1 << a) for a in range(3) if d[a] > 0)]
    variants = (("median split (uploaded tree)", lambda d, t=emit(median_children, items): t), ("LBVH, 30-bit codes (device)", lambda d, t=emit(lbvh_children(), items): t),
                ("binned SAH", lambda d: sah), ("binned SAH x 8 octant orders", pick))
    print("trees built in %.0f s" % (time.time() - t0))
    def bounce(rays):
        out = []
        for o, d in rays:
            _, tt, h = walk(sah, o, d)
            if h >= 0:
                p = o + tt * d; nrm = (p - cen[h]) / rad[h]
                r = rng.normal(size=3); r /= np.linalg.norm(r)
                out.append((p + 1e-3 * nrm, r if r @ nrm > 0 else -r))
        return out
    prim_rays = [(eye, d) for d in D]; b1 = bounce(prim_rays); b2 = bounce(b1)
    sun = np.array([0.3, 0.8, 0.5]); sun /= np.linalg.norm(sun)
    print("rays: %d primary, %d + %d bounce, %d shadow" % (len(prim_rays), len(b1), len(b2), len(b1)))
    for name, f in variants:
        v = [np.mean([walk(f(d), o, d)[0] for o, d in rays]) for rays in (prim_rays, b1, b2)]
        sh = np.mean([walk(f(sun), o, sun, anyhit=True)[0] for o, _ in b1])
        print("%-32s node visits per ray: primary %5.1f  bounce %5.1f  second bounce %5.1f  shadow (any hit) %5.1f" % (name, v[0], v[1], v[2], sh))
    v = [np.mean([walk_stack(sah, o, d)[0] for o, d in rays]) for rays in (prim_rays, b1, b2)]
    sh = np.mean([walk_stack(sah, o, sun, anyhit=True)[0] for o, _ in b1])
    print("%-32s box tests per ray:   primary %5.1f  bounce %5.1f  second bounce %5.1f  shadow (any hit) %5.1f" % ("binned SAH, stack, near first", v[0], v[1], v[2], sh))


if __name__ == "__main__":
    main()
