"""One-GPU PROJECTION of the N-GPU scaling of a BASELINE config: every rank tile of N = 1, 2, 4, 8 (strips=(N, i): 8-row strips dealt
round-robin, what bench.py gives rank i) rendered on the one GPU, kernel time of both launches per tile (HIP events, production
kernels, frames enqueued back to back).  speed-up(N) = t(1) / max_i t(N, i): what N GPUs would reach if nothing but the kernels
mattered (no gather, no launch skew).  A projection, not a measurement of N GPUs.
   python tools/tile_projection.py --config 4 [--spp 64] [--frames 2] --out profiles/r03_tile_scaling_config4.json"""
import sys, os, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--out", default="")
args = ap.parse_args()
cfg = scenes.CONFIGS[args.config]
r = engine.RTRenderer([0])
s = engine.Scene(); scenes.build(args.config, s); r.commit(s)
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=args.spp or None)
res = {"config": cfg.name, "width": p.width, "height": p.height, "spp": p.spp, "frames_per_tile": args.frames, "what": __doc__.split("\n   python")[0], "tiles": {}}
for n in (1, 2, 4, 8):
    ms = []
    for i in range(n):
        r.render_params(p, None, strips=(n, i))
        for _ in range(args.frames):
            r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=(n, i))
        st = r.synchronize()
        ms.append((st.kernel_ms[0] + st.kernel_ms[1]) / st.frames)
    res["tiles"][str(n)] = {"ms_per_rank": [round(v, 3) for v in ms], "max_ms": round(max(ms), 3), "mean_ms": round(sum(ms) / n, 3),
                            "imbalance_max_over_mean": round(max(ms) / (sum(ms) / n), 4)}
    print("N=%d  per-rank ms %s" % (n, " ".join("%.2f" % v for v in ms)), flush=True)
t1 = res["tiles"]["1"]["max_ms"]
res["projected_speedup"] = {n: round(t1 / res["tiles"][n]["max_ms"], 3) for n in ("2", "4", "8")}
res["projected_efficiency"] = {n: round(t1 / res["tiles"][n]["max_ms"] / int(n), 4) for n in ("2", "4", "8")}
print("projected speed-up", res["projected_speedup"], "efficiency", res["projected_efficiency"])
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
