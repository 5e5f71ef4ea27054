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

Rank 0 prints ONE JSON line (schema: task contract) with these extra objects:
  roofline     -- of the path-trace launch (the dominant kernel; for streamed scenes all kernels of the stage).
                  These launches are VALU-bound, not HBM-bound: bound = "valu", achieved = VALU wave-instructions
                  per launch (PMC SQ_INSTS_VALU) / launch time measured live with HIP events on the launch stream,
                  peak = 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction, frac = achieved / peak (<= 1).
                  `hbm` = the measured HBM picture of the same launch: PMC FETCH_SIZE + WRITE_SIZE bytes / launch
                  time against 8 TB/s, plus the compulsory bytes (per-pixel arrays the launch must read and write).
                  `traffic` = those measured bytes.  `algorithmic_bytes_per_launch` (every node /
                  primitive fetch of the reference algorithm priced as memory traffic) is informational only.
                  `useful_lane_frac` = frac x lane_utilisation (issue slots x live lanes); `frac_issue_cycles` weights the
                  quarter-rate instructions (PMC SQ_INSTS_VALU_TRANS_F32) with the 8 cycles they hold a SIMD.
                  PMC counters are collected IN THIS RUN by `rocprofv3 --pmc ... --kernel-trace` passes over
                  `python bench.py --pmc-child` (`pmc_source` says so), or, at N = 1 when rocprofv3 is not usable, read from
                  the committed profiles/r03_pmc_config<N>.json (`pmc_source.kind = "file"`).  At N > 1 rank 0 profiles its
                  OWN tile and prices it against ONE GPU's peak (never single-GPU counts against N x the peak); without
                  rocprofv3 frac is null there.  The extras (configs 3 / 4 / 5) carry a roofline of their own (N = 1; the
                  4K configs' PMC child renders 8 spp and the counts are scaled to the timed sample count).
  cpu_baseline -- the CPU oracle ("port": this repo's restatement of the reference kernels; the C#
                  reference cannot be built here) timed on the host cores over a bounded sample of
                  the same workload (N=1, rank 0 only).
  extra.config3 / config4_full_spp / config5_full_spp -- the other BASELINE configs at their stated size and
                  sample count through the same code (fewer steps), each with its own rays/s, ms and CPU baseline,
                  so the north_star targets (>= 10x CPU on the 10k-sphere scene; scaling on the 4K/100k-triangle
                  config) are visible in the driver-run line.
"""
import argparse
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 1024                  # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4                # max clock (spec); the clock held under load is lower, so frac is conservative
VALU_PEAK_GINST = N_SIMD * CLOCK_GHZ / 2.0     # wave64 VALU instructions per ns over the chip: one per 2 cycles per SIMD-32
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
# optional fourth pass: quarter-rate (transcendental unit) instructions -- v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos, which the
# IEEE division and square-root sequences of these kernels are built around.  A wave64 one occupies its SIMD for 8 cycles instead of 2.
PMC_TRANS = ("SQ_INSTS_VALU_TRANS_F32",)
# sample counts of the PMC child for the 4K configs (their counters are linear in spp: every sample batch does the same work;
# the roofline scales the counts to the timed frame's spp and says so)
PMC_CHILD_SPP = {4: 8, 5: 8}


def algorithmic_bytes(c, n_pixels, launch):
    """ALGORITHMIC bytes of one launch (DESIGN.md 'Measurement') from its work counters."""
    fixed = 48 if launch == 0 else (64 + 12 + 44)
    return (n_pixels * fixed + c["node_visits"] * 44 + c["sphere_tests"] * 84 + c["tri_tests"] * 52 + c["tri_mt_hits"] * 48
            + c["tri_accepted"] * 36 + c["leaf_instances"] * 148 + c["reuse_imports"] * 72)


def sources_sha1():
    """Hash of the kernel sources: PMC instruction counts are a property of the built kernels."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "ilgpu_raytracing_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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
    """Times the oracle on evenly spaced row bands of the same frame until ~budget_s of wall time."""
    from ilgpu_raytracing_amd import _types as T, scenes
    from oracle import orc
    orc.build()
    so = orc.OrcScene()
    scenes.build(cfg_id, so)
    p = scenes.frame_params(cfg, orc.camera_lookat, orc.camera_bake, orc.sun_dir)
    w, h = cfg.width, cfg.height
    arrs, o = T.alloc_outputs(w, h, names=["color", "depth", "objectId", "gb_worldPos", "gb_normalWS", "gb_baseColor", "gb_matId", "gb_objId", "gb_hitMask"])
    threads = host_threads()
    # rows per timed band: >= a few chunks of 128 pixels per thread; smaller bands for the heavy 4K / high-spp frames so
    # that the sample is spread over the image before the budget runs out
    band = 24 if cfg.spp * w <= 16 * 1920 else 8
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


# ------------------------------------------------------------------ PMC counters of the timed kernels
def pmc_child_frames(cfg_id):
    """Production frames `bench.py --pmc-child` renders (what its per-kernel counter sums are divided by)."""
    from ilgpu_raytracing_amd import scenes
    cfg = scenes.CONFIGS[cfg_id]
    return 1 + (3 if cfg.spp * cfg.width <= 16 * 1920 else 1)


def pmc_child(cfg_id, spp, strips=None):
    """The program the rocprofv3 passes run: the same production frames as the timed region (no torch, no counting frame).
    strips = (n, i): the tile of rank i of n (what that rank renders in an N-GPU run)."""
    from ilgpu_raytracing_amd import _types as T, engine, scenes
    cfg = scenes.CONFIGS[cfg_id]
    r = engine.RTRenderer([0])
    s = engine.Scene()
    scenes.build(cfg_id, s)
    r.commit(s)
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=spp or None)
    for _ in range(pmc_child_frames(cfg_id)):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=strips)
    r.synchronize()
    r.close()


def measure_pmc(cfg_id, spp, out_dir, timeout_s=240, strips=None):
    """Three separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md) over
    `bench.py --pmc-child`; kernel trace only.  Returns per-kernel per-frame sums, or None."""
    import collections
    import csv
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    os.makedirs(out_dir, exist_ok=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))        # kernel -> counter -> sum over dispatches
    disp = collections.defaultdict(int)
    for i, ctrs in enumerate(PMC_PASSES + (PMC_TRANS,)):
        optional = i >= len(PMC_PASSES)
        d = os.path.join(out_dir, "pass%d" % i)
        shutil.rmtree(d, ignore_errors=True)
        cmd = [rocprof, "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                                 sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", str(cfg_id)] + (["--spp", str(spp)] if spp else []) \
            + (["--pmc-strips", "%d,%d" % strips] if strips else [])
        try:
            subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=timeout_s)
        except Exception:
            if optional:
                break
            return None
        rows = 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "hrt_" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                rows += 1
                if i == 0:
                    disp[k] += 1
        if rows == 0:
            if optional:
                break                   # this rocprofv3 does not know the counter: frac stays unweighted and says so
            return None
    frames = pmc_child_frames(cfg_id)              # (not counted from dispatches: a frame of the fused kernel can be one launch or two)
    kernels = {}
    for k, cs in per.items():
        kernels[k] = {c: v / frames for c, v in cs.items()}
        kernels[k]["dispatches_per_frame"] = disp.get(k, 0) / frames
        # FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 FETCH_SIZE under-reports 16-B-per-lane coalesced streams by 2x and is
        # uncalibrated for other widths (MI355X_MICROARCH.md 'HBM'); these kernels read 4 / 12 / 16 bytes per lane.
        # Calibrated on the fused kernel's compulsory G-buffer read (profiles/README.md): factor 1.0.
        kernels[k]["hbm_read_bytes"] = kernels[k].get("FETCH_SIZE", 0.0) * 1024.0
        kernels[k]["hbm_write_bytes"] = kernels[k].get("WRITE_SIZE", 0.0) * 1024.0
    return {"frames": frames, "kernels": kernels}


def is_counting_kernel(name):
    """True for the work-counter instantiation of a kernel (its frames are not the timed ones).  COUNT is the first template argument
    of hrt_wf_shade_kernel and the second of every other kernel of the path-trace stage."""
    lt = name.find("<")
    if lt < 0:
        return False
    args, depth, cur = [], 0, ""
    for ch in name[lt + 1:name.rfind(">")]:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip()); cur = ""
        else:
            cur += ch
    args.append(cur.strip())
    pos = 0 if name.startswith("hrt_wf_shade_kernel") else 1
    return len(args) > pos and args[pos] == "true"


def pmc_for(cfg_id, spp, allow_run, out_dir, strips=None, allow_file=True):
    """(per-launch counter sums of the path-trace stage, source description).  Measured now if allowed, else the committed file."""
    src = None
    # N > 1: the other ranks wait for rank 0 in init_process_group meanwhile -- bound the passes well below its time-out
    res = measure_pmc(cfg_id, spp, out_dir, strips=strips, timeout_s=60 if strips else 240) if allow_run else None
    if res is not None:
        src = {"kind": "measured in this run", "how": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*, GRBM_GUI_ACTIVE | SQ_INSTS_VALU_TRANS_F32) + --kernel-trace over `bench.py --pmc-child --config %d`%s%s"
                      % (cfg_id, " --spp %d" % spp if spp else "", " --pmc-strips %d,%d" % strips if strips else ""),
               "frames_profiled": res["frames"]}
        if strips:
            src["tile"] = "strips %d of %d: the tile ONE rank renders, priced against ONE GPU's peak" % (strips[1], strips[0])
    elif allow_file and not strips:
        path = os.path.join(ROOT, "profiles", "r03_pmc_config%d.json" % cfg_id)
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                if (j.get("spp") or None) == (spp or None):
                    res = j
                    src = {"kind": "file", "file": "profiles/" + os.path.basename(path), "sources_sha1_of_file": j.get("sources_sha1"),
                           "sources_sha1_match": j.get("sources_sha1") == sources_sha1()}
            except Exception:
                res = None
    if res is None:
        return None, {"kind": "none"}
    src["child_spp"] = spp or None
    stage = {}
    names = []
    for k, cs in res["kernels"].items():
        if "primary" in k or is_counting_kernel(k):
            continue
        names.append(k)
        for c, v in cs.items():
            stage[c] = stage.get(c, 0.0) + v
    stage["kernels"] = sorted(names)
    res["sources_sha1"] = sources_sha1() if src["kind"] != "file" else res.get("sources_sha1")
    return {"stage": stage, "raw": res}, src


# ------------------------------------------------------------------ one configuration on this rank's strips
class Runner:
    def __init__(self, args, torch, dist, rank, world, dev_index, rehearsal):
        self.args, self.torch, self.dist = args, torch, dist
        self.rank, self.world, self.dev_index, self.rehearsal = rank, world, dev_index, rehearsal

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, sums, maxs):
        torch, dist = self.torch, self.dist
        dev = "cpu" if self.rehearsal else "cuda"
        a = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=dev)
        b = torch.tensor([float(x) for x in maxs], dtype=torch.float64, device=dev)
        if self.world > 1:
            dist.all_reduce(a, op=dist.ReduceOp.SUM)
            dist.all_reduce(b, op=dist.ReduceOp.MAX)
        return a.tolist(), b.tolist()

    def run(self, cfg_id, steps, warmup, spp=0, device_tlas=False, d2h=True):
        """Counting frame + warm-up + K timed frames of config cfg_id on this rank's strips.  Returns a dict (all ranks)."""
        import numpy as np
        from ilgpu_raytracing_amd import _types as T, engine, scenes
        cfg = scenes.CONFIGS[cfg_id]
        if spp:
            cfg = scenes.Config(cfg.name + "_spp%d" % spp, cfg.width, cfg.height, spp, cfg.cam_origin, cfg.cam_lookat,
                                cfg.max_depth, cfg.vfov, cfg.description, cfg.extra)
        rank, world = self.rank, self.world
        r = engine.RTRenderer([self.dev_index])
        try:
            s = engine.Scene()
            scenes.build(cfg_id, s)
            r.commit(s)
            p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
            strips = (world, rank)
            same_picture = None
            if device_tlas:
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
            n_nodes = len(s.arrays()["tlasNodes"]) + len(s.arrays()["blasNodes"])
            fused = n_nodes <= 256            # the library's own choice (hrt_runtime.hip kSmallSceneNodes); reported, not forced

            # untimed counting frame: rays + work counters of this rank's strips (deterministic)
            st = r.render_params(p, None, flags=T.FLAG_COUNTERS, strips=strips)
            c0, c1 = st.k[0].as_dict(), st.k[1].as_dict()
            my_rays = c0["rays_closest"] + c1["rays_closest"] + c1["rays_shadow"]
            my_rows = sum(min(8, cfg.height - sidx * 8) for sidx in range(rank, (cfg.height + 7) // 8, world))
            my_pixels = my_rows * cfg.width
            my_alg = algorithmic_bytes(c1, my_pixels, 1)

            for _ in range(warmup):
                r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=strips)
            r.synchronize()

            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=strips)
            stt = r.synchronize()           # blocks until the K frames are done on this rank's stream
            self.barrier()
            dt = time.perf_counter() - t0
            ft = r.frame_times(0) + r.frame_times(1)                  # per-frame HIP-event time of both launches (last <= 128 frames)
            frames = max(1, stt.frames)
            path_ms, prim_ms = stt.kernel_ms[1] / frames, stt.kernel_ms[0] / frames

            dt_d2h = dt_shared = 0.0
            if d2h:
                # D2H-inclusive variants (reported only, never `value`): blocking frames that gather color/depth/objectId
                # (a) into this rank's own pageable arrays, (b) into ONE page-locked host framebuffer shared by all ranks
                arrs, o = T.alloc_outputs(cfg.width, cfg.height, names=["color", "depth", "objectId"])
                r.render_params(p, o, strips=strips)
                t1 = time.perf_counter()
                nd2h = max(1, min(5, steps))
                for _ in range(nd2h):
                    r.render_params(p, o, strips=strips)
                dt_d2h = (time.perf_counter() - t1) / nd2h
                dt_shared = self.shared_framebuffer_step(r, p, cfg, strips, nd2h)
            sums, maxs = self.reduce([my_rays, my_alg, my_pixels], [dt, path_ms, prim_ms, dt_d2h, dt_shared])
            return {"cfg": cfg, "rays": sums[0], "alg_bytes": sums[1], "pixels": sums[2], "dt": maxs[0], "path_ms": maxs[1], "prim_ms": maxs[2],
                    "d2h_step": maxs[3], "shared_step": maxs[4], "fused": fused, "steps": steps, "same_picture": same_picture,
                    "frame_ms_min": float(np.min(ft)) if len(ft) else None, "frame_ms_median": float(np.median(ft)) if len(ft) else None,
                    "c1": c1}
        finally:
            r.close()

    def shared_framebuffer_step(self, r, p, cfg, strips, n):
        """End-to-end step with every rank gathering its strips into ONE host framebuffer (the /dev/shm arrays of
        tiling.SharedFramebuffer, page-locked through hrt_host_register): barrier, frame + gather, barrier."""
        from ilgpu_raytracing_amd import tiling
        names = ["color", "depth", "objectId"]
        tag = "bench%s" % os.environ.get("MASTER_PORT", str(os.getppid()))
        fb = None
        try:
            if self.rank == 0:
                fb = tiling.SharedFramebuffer(tag, cfg.width, cfg.height, names, create=True)
            if self.world > 1:
                self.dist.barrier()
            if self.rank != 0:
                fb = tiling.SharedFramebuffer(tag, cfg.width, cfg.height, names, create=False)
            try:
                r.register_host(fb.arrays)
                pinned = True
            except Exception:
                pinned = False
            o = fb.outputs_struct()
            r.render_params(p, o, strips=strips)
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                r.render_params(p, o, strips=strips)
                if self.world > 1:
                    self.dist.barrier()
            dt = (time.perf_counter() - t0) / n
            if pinned:
                r.unregister_host(fb.arrays)
            return dt
        except Exception:
            return 0.0
        finally:
            if self.world > 1:
                self.dist.barrier()
            if fb is not None:
                fb.close()


def roofline_of(res, pmc, pmc_src, world, count_scale=1.0, tile_of=1):
    """VALU-issue roofline of the path-trace stage + its measured HBM picture (see the module docstring).
    world: GPUs whose peak the counts are priced against (1 when the counts are those of one rank's tile);
    count_scale: timed spp / spp of the PMC child (counters are linear in the sample count);
    tile_of: N when the counts are those of ONE rank's tile of an N-rank frame (compulsory and algorithmic bytes are then the tile's)."""
    cfg = res["cfg"]
    kernel = ("hrt_path_trace_kernel<TracerFlat> (fused; sample-group split kernel + resolve on small tiles)" if res["fused"]
              else "path-trace stage, streamed: hrt_wf_{shade,walk_shadow,walk_closest,finish,resolve}_kernel")
    launch_s = res["path_ms"] * 1e-3
    compulsory = res["pixels"] * (48 + 12 + 44) / float(tile_of)      # G-buffer read 48 B + framebuffer write 12 B + reservoir write <= 44 B per pixel
    out = {"bound": "valu", "kernel": kernel, "launch_ms": round(res["path_ms"], 4),
           "unit": "G wave64 VALU instructions / s",
           "peak": round(VALU_PEAK_GINST * world, 1),
           "peak_is": "%d SIMD-32 x %.1f GHz / 2 cycles per wave64 VALU instruction, x %d GPU(s)" % (N_SIMD, CLOCK_GHZ, world),
           "achieved": None, "frac": None, "traffic": None,
           "algorithmic_bytes_per_launch": int(res["alg_bytes"] / float(tile_of)),
           "algorithmic_note": "SURVEY 8d figure (reference struct sizes x work counters: every node / primitive fetch priced as memory traffic); "
                               "informational only -- the trees are cache-resident, this is not HBM traffic and is never used as frac",
           "pmc_source": pmc_src}
    if tile_of > 1:
        out["priced"] = "rank 0's tile (1/%d of the frame, interleaved 8-row strips) against ONE GPU's peak over the slowest rank's launch time" % tile_of
    if pmc is not None and launch_s > 0:
        stg = dict(pmc["stage"])
        if count_scale != 1.0:
            for k_, v_ in list(stg.items()):
                if isinstance(v_, float) and k_ != "dispatches_per_frame":
                    stg[k_] = v_ * count_scale
            out["counts_scaled_by"] = round(count_scale, 4)
        insts = stg.get("SQ_INSTS_VALU", 0.0)
        achieved = insts / launch_s / 1e9
        out["achieved"] = round(achieved, 2)
        out["frac"] = round(achieved / (VALU_PEAK_GINST * world), 4)
        out["valu_insts_per_launch"] = int(insts)
        if stg.get("SQ_ACTIVE_INST_VALU"):
            out["lane_utilisation"] = round(stg.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * stg["SQ_ACTIVE_INST_VALU"]), 4)
            out["useful_lane_frac"] = round(out["frac"] * out["lane_utilisation"], 4)       # issue slots x live lanes: what frac hides
        # issue-CYCLE occupancy: a quarter-rate instruction holds its SIMD four times as long as the others
        trans = stg.get("SQ_INSTS_VALU_TRANS_F32")
        if trans is not None and insts > 0:
            out["trans_insts_per_launch"] = int(trans)
            out["frac_issue_cycles"] = round((insts + 3.0 * trans) / launch_s / 1e9 / (VALU_PEAK_GINST * world), 4)
            out["frac_issue_cycles_is"] = "(VALU instructions + 3 x quarter-rate ones) x 2 cycles / (SIMD-cycles of the launch): v_rcp / v_rsq / v_sqrt ... occupy a SIMD for 8 cycles per wave64"
        else:
            out["frac_issue_cycles"] = None
            out["frac_issue_cycles_is"] = "not measured: this rocprofv3 has no SQ_INSTS_VALU_TRANS_F32; frac prices every instruction at the full rate (2 cycles)"
        if stg.get("SQ_WAVE_CYCLES"):
            out["wave_time_waiting"] = round(stg.get("SQ_WAIT_ANY", 0.0) / stg["SQ_WAVE_CYCLES"], 4)
        traffic = stg.get("hbm_read_bytes", 0.0) + stg.get("hbm_write_bytes", 0.0)
        out["traffic"] = int(traffic)
        gbs = traffic / launch_s / 1e9
        out["hbm"] = {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(gbs / (HBM_PEAK_GBS * world), 5),
                      "read_bytes": int(stg.get("hbm_read_bytes", 0.0)), "write_bytes": int(stg.get("hbm_write_bytes", 0.0)),
                      "compulsory_bytes": int(compulsory), "compulsory_frac_of_peak": round(compulsory / launch_s / 1e9 / (HBM_PEAK_GBS * world), 5),
                      "traffic_over_compulsory": round(traffic / max(1.0, compulsory), 2)}
        out["kernels_priced"] = stg.get("kernels")
    else:
        out["note"] = "PMC counters unavailable for this run (no rocprofv3 and no committed profile): frac not computed"
    return out


def summary_of(res, world, cpu=None):
    rays = res["rays"]
    o = {"workload": res["cfg"].name, "width": res["cfg"].width, "height": res["cfg"].height, "spp": res["cfg"].spp,
         "steps": res["steps"], "rays_per_step": int(rays),
         "mrays_per_s": round(rays * res["steps"] / res["dt"] / 1e6, 2), "ms_per_step": round(res["dt"] / res["steps"] * 1e3, 4),
         "primary_kernel_ms": round(res["prim_ms"], 4), "path_trace_stage_ms": round(res["path_ms"], 4),
         "frame_event_ms_min": res["frame_ms_min"], "frame_event_ms_median": res["frame_ms_median"],
         "organisation": "fused" if res["fused"] else "streamed"}
    if cpu is not None:
        o["cpu_baseline"] = cpu
        o["ratio_vs_cpu_baseline"] = round(o["mrays_per_s"] / cpu["value"], 1) if cpu["value"] > 0 else None
    return o


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
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto", help="auto: collect the roofline's PMC counters with rocprofv3 passes in this run (N = 1)")
    ap.add_argument("--extras", type=int, default=1, help="1: also run BASELINE configs 3, 4 and 5 at their stated size and spp (fewer steps) into `extra`")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-strips", default="", help=argparse.SUPPRESS)
    ap.add_argument("--save-pmc", default=None, help="write the measured per-kernel PMC sums to this JSON (profiles/r02_pmc_config<N>.json)")
    args = ap.parse_args()

    if args.pmc_child:
        pmc_child(args.config, args.spp, tuple(int(v) for v in args.pmc_strips.split(",")) if args.pmc_strips else None)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    # PMC passes first: child processes, before this process touches the GPU.
    # N = 1: the whole frame, measured now (or the committed profile when rocprofv3 is unusable).
    # N > 1: rank 0 profiles ITS OWN tile (strips (N, 0): interleaved 8-row strips, every rank's tile is statistically the same; small
    #        tiles run the sample-group split kernel, which is what gets counted) and prices it against ONE GPU's peak over the slowest
    #        rank's launch time.  Never the N = 1 counts against N x the peak.
    pmc = None
    pmc_src = {"kind": "none"}
    extra_pmc = {}
    if rank == 0:
        have_gpu = os.path.exists("/dev/kfd")
        run_ok = args.pmc == "auto" and have_gpu
        pmc_dir = os.path.join(ROOT, "gpurun_out", "bench_pmc_config%d%s" % (args.config, "_tile%d" % world if world > 1 else ""))
        child_spp = args.spp or PMC_CHILD_SPP.get(args.config, 0)
        pmc, pmc_src = pmc_for(args.config, child_spp, run_ok, pmc_dir, strips=(world, 0) if world > 1 else None, allow_file=world == 1)
        if world > 1 and pmc is None:
            pmc_src = {"kind": "none", "reason": "N > 1 and no in-run PMC of rank 0's tile (rocprofv3 unusable): frac is not computed rather than priced from single-GPU counts"}
        if args.save_pmc and pmc is not None and pmc_src.get("kind") != "file":
            raw = dict(pmc["raw"], config=args.config, spp=child_spp or None, sources_sha1=sources_sha1(),
                       note="per-kernel sums per frame; FETCH_SIZE / WRITE_SIZE in KiB (hbm_*_bytes = x1024); rocprofv3 --pmc passes of `bench.py --pmc-child`")
            json.dump(raw, open(args.save_pmc, "w"), indent=1, sort_keys=True)
        if world == 1 and run_ok and args.extras and args.config == 2 and not args.spp and not args.device_tlas:
            for cid in (3, 4, 5):          # the extras get their own roofline (4K configs: PMC child at 8 spp, counts scaled to the timed spp)
                t_p = time.perf_counter()
                pm, src = pmc_for(cid, PMC_CHILD_SPP.get(cid, 0), True, os.path.join(ROOT, "gpurun_out", "bench_pmc_config%d" % cid), allow_file=False)
                src["seconds"] = round(time.perf_counter() - t_p, 1)
                extra_pmc[cid] = (pm, src)
                if args.save_pmc and pm is not None:
                    raw = dict(pm["raw"], config=cid, spp=PMC_CHILD_SPP.get(cid) or None, sources_sha1=sources_sha1(),
                               note="per-kernel sums per frame; FETCH_SIZE / WRITE_SIZE in KiB (hbm_*_bytes = x1024); rocprofv3 --pmc passes of `bench.py --pmc-child`")
                    json.dump(raw, open(args.save_pmc.replace("config%d" % args.config, "config%d" % cid) if "config%d" % args.config in args.save_pmc else args.save_pmc + ".config%d" % cid, "w"), indent=1, sort_keys=True)

    import torch            # loaded first: its bundled HIP runtime is the one the process uses
    import torch.distributed as dist

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

    R = Runner(args, torch, dist, rank, world, dev_index, rehearsal)
    main_res = R.run(args.config, args.steps, args.warmup, spp=args.spp, device_tlas=args.device_tlas)

    extras = {}
    if args.extras and args.config == 2 and not args.spp and not args.device_tlas:
        # the other BASELINE configs at their stated size and sample count: fewer steps, same code path
        for name, cid, k, w_ in (("config3", 3, min(args.steps, 20), min(args.warmup, 3)), ("config4_full_spp", 4, min(args.steps, 3), 1),
                                 ("config5_full_spp", 5, min(args.steps, 2), 1)):
            try:
                extras[name] = R.run(cid, max(1, k), w_, d2h=False)
            except Exception as e:             # an extra must never take the headline line down
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        cfg = main_res["cfg"]
        value = main_res["rays"] * args.steps / main_res["dt"] / 1e6
        out = {
            "metric": "Mrays/sec at %dx%d %dspp" % (cfg.width, cfg.height, cfg.spp),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(main_res["dt"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if rehearsal else ""),
            "config": {"workload": cfg.name, "description": cfg.description, "width": cfg.width, "height": cfg.height, "spp": cfg.spp,
                       "max_depth": cfg.max_depth, "frame": 0, "restir_reuse": False,
                       "tlas": ("rebuilt on the device (LBVH); picture bit-identical to the uploaded tree's: %s" % main_res["same_picture"]) if args.device_tlas else "as uploaded (reference builder)",
                       "parallelism": "row-strips x%d (8-row strips, round-robin)" % world, "rays_per_step": int(main_res["rays"])},
            "roofline": roofline_of(main_res, pmc, pmc_src, 1, tile_of=world,
                                    count_scale=(cfg.spp / float(pmc_src["child_spp"])) if pmc_src.get("child_spp") else 1.0),
            "extra": {"primary_kernel_ms": round(main_res["prim_ms"], 4), "path_trace_kernel_ms": round(main_res["path_ms"], 4),
                      "frame_event_ms_min": main_res["frame_ms_min"], "frame_event_ms_median": main_res["frame_ms_median"],
                      "step_ms_with_d2h_gather": round(main_res["d2h_step"] * 1e3, 4),
                      "mrays_per_s_with_d2h_gather": round(main_res["rays"] / main_res["d2h_step"] / 1e6, 2) if main_res["d2h_step"] > 0 else None,
                      "step_ms_shared_pinned_framebuffer": round(main_res["shared_step"] * 1e3, 4),
                      "mrays_per_s_shared_pinned_framebuffer": round(main_res["rays"] / main_res["shared_step"] / 1e6, 2) if main_res["shared_step"] > 0 else None},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.config, cfg, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        for name, res in extras.items():
            if "error" in res:
                out["extra"][name] = res
                continue
            cid = {"config3": 3, "config4_full_spp": 4, "config5_full_spp": 5}[name]
            cpu = cpu_baseline(cid, res["cfg"], min(6.0, args.cpu_seconds)) if (world == 1 and args.cpu_seconds > 0) else None
            out["extra"][name] = summary_of(res, world, cpu)
            if cid in extra_pmc:
                pm, src = extra_pmc[cid]
                out["extra"][name]["roofline"] = roofline_of(res, pm, src, 1, count_scale=(res["cfg"].spp / float(src["child_spp"])) if src.get("child_spp") else 1.0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
