#!/usr/bin/env python3
"""Wall time of hrt_scene_upload (engine.RTRenderer.commit) for scenes of N one-sphere instances: what the second tree (host SAH
topology, four numberings) adds to an upload.   python tools/upload_time.py [--counts 1000,10000,100000,300000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import engine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--counts", default="1000,10000,100000,300000")
args = ap.parse_args()
r = engine.RTRenderer([0])
for n in [int(x) for x in args.counts.split(",")]:
    t = time.perf_counter(); s = engine.Scene(); scenes.build_random_spheres(s, n, extent=20.0 * (n / 10000.0) ** 0.5); tb = time.perf_counter() - t
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); r.commit(s); best = min(best, time.perf_counter() - t)
    print("instances %7d: host scene build %.3f s, upload (validate, pack, copy, second tree) %.3f s" % (n + 1, tb, best), flush=True)
