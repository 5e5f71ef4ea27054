"""Deforming meshes: device-side BLAS refit (hrt_scene_update_positions) against the reference's way (host BLAS + TLAS rebuild
and re-upload of all arrays, Scene.cs:405-467 / :258-279), and the frame time before / after.
   python tools/mesh_update_bench.py [--configs 4,5] [--frames 5] [--out profiles/x.json]"""
import sys, os, time, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="4,5")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--spp", type=int, default=2)
ap.add_argument("--out", default="")
args = ap.parse_args()
r = engine.RTRenderer([0])


def frame_ms(p):
    r.render_params(p, None)
    for _ in range(args.frames):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC)
    st = r.synchronize()
    return (st.kernel_ms[0] + st.kernel_ms[1]) / st.frames


rows = []
for cid in [int(c) for c in args.configs.split(",")]:
    cfg = scenes.CONFIGS[cid]
    t = time.perf_counter(); s = engine.Scene(); scenes.build(cid, s); t_build = time.perf_counter() - t
    t = time.perf_counter(); r.commit(s); t_up = time.perf_counter() - t
    a = s.arrays()
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=args.spp)
    row = {"config": cid, "triangles": int(len(a["meshTris"])), "blas_nodes": int(len(a["blasNodes"])), "vertices": int(len(a["meshPositions"])),
           "host_scene_build_ms": t_build * 1e3, "host_upload_all_ms": t_up * 1e3, "frame_ms_before": frame_ms(p)}
    pos = np.stack([a["meshPositions"][f] for f in "XYZ"], axis=1)
    new = (pos * (1.0 + 0.03 * np.sin(5.0 * pos[:, [1, 2, 0]]))).astype(np.float32)
    best, st = 1e9, None
    for _ in range(5):
        t = time.perf_counter(); st = r.update_positions(0, new, T.REBUILD_FORCE_REFIT); best = min(best, time.perf_counter() - t)
    row["device_refit_ms_wall_with_h2d"], row["device_refit_ms_kernels"] = best * 1e3, st.device_ms
    row["frame_ms_after_refit"] = frame_ms(p)
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); st = r.update_positions(0, new, T.REBUILD_FORCE_REFIT | T.REBUILD_BLAS); best = min(best, time.perf_counter() - t)
    row["device_blas_rebuild_ms_wall_with_h2d"], row["device_blas_rebuild_ms_kernels"] = best * 1e3, st.device_ms
    row["frame_ms_after_blas_rebuild"] = frame_ms(p)
    # the undeformed mesh under a device-built BLAS: tree quality of the LBVH against the reference's median split
    r.update_positions(0, pos, T.REBUILD_FORCE_REFIT | T.REBUILD_BLAS)
    row["frame_ms_undeformed_device_blas"] = frame_ms(p)
    rows.append(row)
    print(json.dumps(row), flush=True)
if args.out:
    with open(args.out, "w") as f:
        json.dump({"tool": "tools/mesh_update_bench.py", "frame": "BASELINE camera of the config, %d spp" % args.spp, "rows": rows}, f, indent=1)
