"""Moving instances: device-side TLAS refit / rebuild (hrt_scene_update_instances) against the reference's way
(host RebuildTLAS + re-upload of all arrays, BvhManager.cs:27 / Scene.cs:258-279,358-368), and what each tree costs a frame.
   python tools/bvh_update_bench.py [--counts 10000,100000] [--frames 5] [--out profiles/x.json]
Scene: ground + N single-sphere instances (config 3 at N = 10000), 1920x1080, 4 spp."""
import sys, os, time, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--counts", default="10000,100000")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--out", default="")
args = ap.parse_args()

r = engine.RTRenderer([0])
cfg = scenes.CONFIGS[3]
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=args.spp)


def frame_ms():
    r.render_params(p, None)
    for _ in range(args.frames):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC)
    st = r.synchronize()
    return (st.kernel_ms[0] + st.kernel_ms[1]) / st.frames


def timed(fn, reps=5):
    best, out = 1e9, None
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t)
    return best * 1e3, out


results = []
for n in [int(x) for x in args.counts.split(",")]:
    s = engine.Scene(); scenes.build_random_spheres(s, n, extent=20.0 * (n / 10000.0) ** 0.5)
    r.commit(s)
    row = {"instances": n + 1, "tlas_nodes_host": int(len(s.arrays()["tlasNodes"]))}
    row["frame_ms_host_tree_static"] = frame_ms()
    # the same static scene, tree rebuilt on the device (fast-sphere path stays on): tree quality alone
    ms, st = timed(lambda: r.update_instances([], [], T.REBUILD_FORCE_REBUILD))
    row["device_rebuild_ms_wall"], row["device_rebuild_ms_kernels"] = ms, st.device_ms
    row["tlas_nodes_device"], row["sah_host_tree"] = st.tlas_nodes, None
    row["sah_device_tree"] = st.sah_cost
    row["frame_ms_device_tree_static"] = frame_ms()
    r.commit(s)
    st = r.update_instances([], [], T.REBUILD_FORCE_REFIT)
    row["sah_host_tree"] = st.sah_cost
    # every sphere jitters (translation): the dynamic-scene step
    ids = np.arange(1, n + 1, dtype=np.int32)
    rng = np.random.default_rng(7)
    xf = np.zeros((n, 12), np.float32); xf[:, 0] = xf[:, 5] = xf[:, 10] = 1.0
    xf[:, [3, 7, 11]] = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32) * np.array([1.0, 0.2, 1.0], np.float32)
    ms, st = timed(lambda: r.update_instances(ids, xf, T.REBUILD_FORCE_REFIT))
    row["device_refit_ms_wall"], row["device_refit_ms_kernels"], row["growth_after_refit"] = ms, st.device_ms, st.growth_refit
    row["frame_ms_moved_refit_tree"] = frame_ms()
    ms, st = timed(lambda: r.update_instances(ids, xf, T.REBUILD_FORCE_REBUILD))
    row["device_move_rebuild_ms_wall"], row["device_move_rebuild_ms_kernels"] = ms, st.device_ms
    row["frame_ms_moved_device_rebuilt_tree"] = frame_ms()
    # the same motion expressed as sphere data (hrt_scene_update_spheres): the instances keep their identity transform and
    # with it the walkers' fast path
    r.commit(s)
    sp = s.arrays()["spheres"].copy()
    for k, f in enumerate("XYZ"):
        sp["center"][f][1:] += xf[:, [3, 7, 11][k]]
    ms, st = timed(lambda: r.update_spheres(1, sp[1:], T.REBUILD_FORCE_REFIT))
    row["device_spheres_refit_ms_wall"], row["device_spheres_refit_ms_kernels"] = ms, st.device_ms
    row["frame_ms_moved_spheres_refit_tree"] = frame_ms()
    ms, st = timed(lambda: r.update_spheres(1, sp[1:], T.REBUILD_FORCE_REBUILD))
    row["device_spheres_rebuild_ms_wall"], row["device_spheres_rebuild_ms_kernels"] = ms, st.device_ms
    row["frame_ms_moved_spheres_device_rebuilt_tree"] = frame_ms()
    # the reference's way: records + RebuildTLAS on the host, then UploadAll
    affs = []
    for k in range(n):
        m = T.identity_affine(); m.m03, m.m13, m.m23 = float(xf[k, 3]), float(xf[k, 7]), float(xf[k, 11]); affs.append(m)

    def host_path():
        for k in range(n):
            s.set_instance_transform(int(ids[k]), affs[k])
        t1 = time.perf_counter(); s.rebuild_tlas(); t2 = time.perf_counter(); r.commit(s); t3 = time.perf_counter()
        return (t2 - t1) * 1e3, (t3 - t2) * 1e3
    ms, (tb, tu) = timed(host_path, reps=2)
    row["host_rebuild_tlas_ms"], row["host_upload_all_ms"] = tb, tu
    row["frame_ms_moved_host_rebuilt_tree"] = frame_ms()
    results.append(row)
    print(json.dumps(row), flush=True)
if args.out:
    with open(args.out, "w") as f:
        json.dump({"tool": "tools/bvh_update_bench.py", "frame": "1920x1080, %d spp, config 3 camera" % args.spp, "rows": results}, f, indent=1)
