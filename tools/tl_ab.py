"""A/B of the treelet-queued walker (hrt_walker_tl.hpp, HRT_FLAG_TREELETS) against the plain persistent-wave walker on the mesh configs:
the same frame without and with the flag, every output array compared bit for bit, kernel time of the path stage.
   python tools/tl_ab.py [--configs 4,5] [--spp 4:4,5:2] [--frames 4] [--small]"""
import sys, os, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="4,5")
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--spp", default="4:4,5:2")
ap.add_argument("--small", action="store_true", help="960x540 instead of the config's size (quick parity check)")
ap.add_argument("--limits", default="", help="bytes,minNodes,minBlasNodes of the treelet cut (default: the shipped values)")
ap.add_argument("--nocheck", action="store_true")
ap.add_argument("--only", default="", help="plain | treelets: run one side only (profiling)")
ap.add_argument("--save", default="", help="with --only: write this side's output arrays to <file>.cfgN.npz")
ap.add_argument("--compare", default="", help="with --only: compare this side's output arrays with <file>.cfgN.npz (another process / build / environment)")
args = ap.parse_args()
spp = dict((int(a), int(b)) for a, b in (x.split(":") for x in args.spp.split(",")))
lim = [int(v) for v in args.limits.split(",")] if args.limits else None
L = engine.hooks() if lim else engine.lib()          # other limits than the shipped ones: the hooks build (include/hrt_test_hooks.h)
if lim: L.hrt_debug_set_treelet_limits(*lim)
names = ["color", "depth", "objectId", "radiance", "res_L", "res_wi", "res_m", "res_w", "res_wSum"]
r = engine.RTRenderer([0], library=L)
for cid in [int(c) for c in args.configs.split(",")]:
    cfg = scenes.CONFIGS[cid]
    s = engine.Scene(); scenes.build(cid, s)
    kw = dict(width=960, height=540) if args.small else {}
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=spp.get(cid, cfg.spp), **kw)
    res = {}
    for label, on in (("plain", False), ("treelets", True)):
        if args.only and args.only != label: continue
        fl = T.FLAG_TREELETS if on else 0
        r.commit(s)
        out, o = T.alloc_outputs(p.width, p.height, names)
        r.reset_history()
        r.render_params(p, o, flags=fl)
        for _ in range(args.frames):
            r.render_params(p, None, flags=fl | T.FLAG_NO_SYNC)
        st = r.synchronize()
        res[label] = (out, st.kernel_ms[0] / st.frames, st.kernel_ms[1] / st.frames)
        print("cfg%d %dx%d spp%d %-8s primary %.3f ms  path stage %.3f ms" % (cid, p.width, p.height, p.spp, label, res[label][1], res[label][2]), flush=True)
    if args.only and (args.save or args.compare):
        mine = res[args.only][0]
        if args.save: np.savez(args.save + ".cfg%d.npz" % cid, **mine)
        if args.compare:
            other = np.load(args.compare + ".cfg%d.npz" % cid)
            bad = [k for k in mine if np.ascontiguousarray(mine[k]).tobytes() != np.ascontiguousarray(other[k]).tobytes()]
            print("   %s vs %s: %s" % (args.only, args.compare, "IDENTICAL" if not bad else "MISMATCH %s" % bad), flush=True)
    if not args.nocheck and not args.only:
        a, b = res["plain"][0], res["treelets"][0]
        bad = {}
        for k in a:
            x, y = np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])
            if x.dtype == np.float32:
                eq = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
            else:
                eq = x == y
            if not eq.all(): bad[k] = int((~eq).sum())
        print("   treelets vs plain: %s   speed-up of the path stage %.3fx" % ("IDENTICAL" if not bad else "MISMATCH %s" % bad, res["plain"][2] / res["treelets"][2]), flush=True)
if lim: L.hrt_debug_set_treelet_limits(0, 0, 0)
