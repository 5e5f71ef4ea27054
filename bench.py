#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render hot path on N MI355X GPUs (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one frame of the hot path: launch 1 (primary visibility) + launch 2 (path trace) over the
whole image of the configuration, scene and all per-pixel buffers resident in HBM, results left in
HBM (the D2H gather of the 12 B/pixel framebuffer is reported separately in `extra`, never in
`value`).  Default workload = BASELINE.json configs[1]: the 8-sphere Cornell-style scene at
1920x1080, 4 spp, maxDepth 3, frame 0, ReSTIR reuse off (single frame).

N > 1: the image is cut into 8-row strips dealt round-robin to the ranks (scene replicated, no
data-path collective; each rank owns its strips of the one frame => total work is fixed as N grows:
"scaling": "strong").  Rays per step come from one untimed counting frame (the work counters are
deterministic).  The K timed steps are enqueued back to back on the rank's HIP stream
(HRT_FLAG_NO_SYNC) and bracketed by barrier + synchronize; value = rays of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (schema: task contract) with two extra objects:
  roofline     -- HBM roofline of the dominant kernel (path trace): achieved = ALGORITHMIC bytes of one
                  launch (reference struct sizes x work counters) / mean launch time measured
                  with HIP events on the launch stream inside the timed region; peak 8 TB/s.
  cpu_baseline -- the CPU oracle ("port": this repo's restatement of the reference kernels; the C#
                  reference cannot be built here) timed on the host cores over a bounded sample of
                  the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(c, n_pixels, launch):
    """ALGORITHMIC bytes of one launch (DESIGN.md 'Measurement') from its work counters."""
    fixed = 48 if launch == 0 else (64 + 12 + 44)
    return (n_pixels * fixed + c["node_visits"] * 44 + c["sphere_tests"] * 84 + c["tri_tests"] * 52 + c["tri_mt_hits"] * 48
            + c["tri_accepted"] * 36 + c["leaf_instances"] * 148 + c["reuse_imports"] * 72)


def host_threads():
    """Threads the oracle may use: the affinity mask capped by the cgroup CPU quota (the GPU box gives a
    1-GPU job a share of the host's cores, not all of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()                      # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:                                                                        # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(cfg_id, cfg, budget_s):
    """Times the oracle on evenly spaced 8-row bands of the same frame until ~budget_s of wall time."""
    import numpy as np
    from ilgpu_raytracing_amd import _types as T, scenes
    from oracle import orc
    orc.build()
    so = orc.OrcScene()
    scenes.build(cfg_id, so)
    p = scenes.frame_params(cfg, orc.camera_lookat, orc.camera_bake, orc.sun_dir)
    w, h = cfg.width, cfg.height
    arrs, o = T.alloc_outputs(w, h, names=["color", "depth", "objectId", "gb_worldPos", "gb_normalWS", "gb_baseColor", "gb_matId", "gb_objId", "gb_hitMask"])
    threads = host_threads()
    band = 24                      # rows per timed band: >= 360 chunks of 128 pixels, keeps every thread busy
    strips = list(range(0, (h + band - 1) // band))
    order = []                   # bit-reversal-like spread: 0, S/2, S/4, 3S/4, ...
    step = len(strips)
    seen = set()
    while step >= 1 and len(order) < len(strips):
        for s in range(0, len(strips), max(1, step)):
            if s not in seen:
                seen.add(s); order.append(s)
        step //= 2
    rays = 0
    rows = 0
    t0 = time.perf_counter()
    for s in order:
        y0, y1 = s * band, min(h, s * band + band)
        st = orc.render_frame(so.desc(), p, o, row_begin=y0, row_end=y1, nthreads=threads)
        rays += sum(st.k[i].rays_closest + st.k[i].rays_shadow for i in range(2))
        rows += y1 - y0
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%d of %d rows (evenly spread %d-row bands) of the same %dx%d %d-spp frame, %.1f s wall on %d threads"
                      % (rows, h, band, w, h, cfg.spp, dt, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json configs[] index + 1 (default 2 = configs[1])")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (non-default => not the headline config)")
    ap.add_argument("--device-tlas", action="store_true", help="after the upload rebuild the TLAS on the device (hrt_scene_update_instances, "
                    "ForceRebuild): same picture, another tree than the reference's builder makes (not the headline setting)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch            # loaded first: its bundled HIP runtime is the one the process uses
    import torch.distributed as dist
    from ilgpu_raytracing_amd import _types as T, engine, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists for the product)")
    # HRT_BENCH_REHEARSAL=1: functional rehearsal of the N-rank path on a 1-GPU box (all ranks share device 0,
    # gloo instead of RCCL).  Never used for reported numbers.
    rehearsal = os.environ.get("HRT_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = scenes.CONFIGS[args.config]
    if args.spp:
        cfg = scenes.Config(cfg.name + "_spp%d" % args.spp, cfg.width, cfg.height, args.spp, cfg.cam_origin, cfg.cam_lookat,
                            cfg.max_depth, cfg.vfov, cfg.description, cfg.extra)
    r = engine.RTRenderer([dev_index])
    s = engine.Scene()
    scenes.build(args.config, s)
    r.commit(s)
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
    strips = (world, rank)
    same_picture = None
    if args.device_tlas:
        # the frame on the uploaded tree and on the device-built one must be the same picture (they are unless two
        # instances are hit at bit-equal distance, or the scene holds rotated / enlarged instances: DESIGN.md 4)
        names = ["color", "depth", "objectId", "radiance"]
        a0, o0 = T.alloc_outputs(cfg.width, cfg.height, names=names)
        r.render_params(p, o0, strips=strips)
        r.update_instances([], [], T.REBUILD_FORCE_REBUILD)
        r.reset_history()
        a1, o1 = T.alloc_outputs(cfg.width, cfg.height, names=names)
        r.render_params(p, o1, strips=strips)
        r.reset_history()
        same_picture = all(a0[k].tobytes() == a1[k].tobytes() for k in names)
    P = cfg.width * cfg.height
    n_nodes = len(s.arrays()["tlasNodes"]) + len(s.arrays()["blasNodes"])
    fused = n_nodes <= 256            # the library's own choice (hrt_runtime.hip kSmallSceneNodes); reported, not forced

    # untimed counting frame: rays + work counters of this rank's strips (deterministic)
    st = r.render_params(p, None, flags=T.FLAG_COUNTERS, strips=strips)
    c0, c1 = st.k[0].as_dict(), st.k[1].as_dict()
    my_rays = c0["rays_closest"] + c1["rays_closest"] + c1["rays_shadow"]
    my_rows = sum(min(8, cfg.height - sidx * 8) for sidx in range(rank, (cfg.height + 7) // 8, world))
    my_pixels = my_rows * cfg.width
    my_bytes = [algorithmic_bytes(c0, my_pixels, 0), algorithmic_bytes(c1, my_pixels, 1)]

    for _ in range(args.warmup):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=strips)
    r.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=strips)
    stt = r.synchronize()           # blocks until the K frames are done on this rank's stream
    barrier()
    dt = time.perf_counter() - t0

    # D2H-inclusive variant (reported only): blocking frames that gather color/depth/objectId to host
    arrs, o = T.alloc_outputs(cfg.width, cfg.height, names=["color", "depth", "objectId"])
    r.render_params(p, o, strips=strips)
    t1 = time.perf_counter()
    nd2h = max(1, min(5, args.steps))
    for _ in range(nd2h):
        std = r.render_params(p, o, strips=strips)
    dt_d2h = (time.perf_counter() - t1) / nd2h

    red_dev = "cpu" if rehearsal else "cuda"
    tot = torch.tensor([float(my_rays), float(my_bytes[1])], dtype=torch.float64, device=red_dev)
    mx = torch.tensor([dt, stt.kernel_ms[1] / max(1, stt.frames), stt.kernel_ms[0] / max(1, stt.frames), dt_d2h], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    rays_total, bytes_total = tot[0].item(), tot[1].item()
    dt_max, path_ms, prim_ms, d2h_step = mx[0].item(), mx[1].item(), mx[2].item(), mx[3].item()

    if rank == 0:
        value = rays_total * args.steps / dt_max / 1e6
        achieved = bytes_total / (path_ms * 1e-3) / 1e9           # GB/s over all ranks' launches (max launch time)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_config%d.json" % args.config)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch") if tj.get("spp") in (None, cfg.spp) else None      # measured at this sample count only
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec at %dx%d %dspp" % (cfg.width, cfg.height, cfg.spp),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if rehearsal else ""),
            "config": {"workload": cfg.name, "description": cfg.description, "width": cfg.width, "height": cfg.height, "spp": cfg.spp,
                       "max_depth": cfg.max_depth, "frame": 0, "restir_reuse": False, "tlas": ("rebuilt on the device (LBVH); picture bit-identical to the uploaded tree's: %s" % same_picture) if args.device_tlas else "as uploaded (reference builder)", "parallelism": "row-strips x%d (8-row strips, round-robin)" % world,
                       "rays_per_step": int(rays_total)},
            "roofline": {"bound": "hbm", "kernel": ("hrt_path_trace_kernel (fused)" if fused else "path-trace stage, streamed: hrt_wf_{init,shade,walk_shadow,walk_closest,finish,resolve}_kernel"), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": round(achieved / (HBM_PEAK_GBS * world), 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_total), "launch_ms": round(path_ms, 4),
                         "note": "algorithmic bytes price every node / instance / primitive fetch of the reference algorithm as memory traffic "
                                 "(SURVEY 8d); the trees are cache-resident, so achieved exceeds the HBM peak: the launch is VALU-bound, "
                                 "measured HBM bytes per launch are in traffic"},
            "extra": {"primary_kernel_ms": round(prim_ms, 4), "path_trace_kernel_ms": round(path_ms, 4),
                      "step_ms_with_d2h_gather": round(d2h_step * 1e3, 4),
                      "mrays_per_s_with_d2h_gather": round(rays_total / d2h_step / 1e6, 2)},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.config, cfg, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
