"""Measured HBM traffic of one path-trace launch (all kernels of the stage) from rocprofv3 PMC passes.

Run on the GPU box:  python tools/collect_traffic.py --config 2 --out profiles/traffic_config2.json
Two separate passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit together), kernel trace only,
over `bench.py --steps 3 --warmup 1 --cpu-seconds 0`.  Corrections per MI355X_MICROARCH.md 'HBM': FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a coalesced read stream -> doubled (checked here
against the known compulsory read of the fused kernel: 48 B/pixel of G-buffer).
"""
import argparse, csv, glob, json, os, subprocess, sys, collections

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--out", default=None)
ap.add_argument("--dir", default="gpurun_out/traffic")
ap.add_argument("--fetch-factor", type=float, default=1.0)
ap.add_argument("--spp", type=int, default=0, help="bench.py --spp override (0: the config's own)")
args = ap.parse_args()
# MI355X_MICROARCH.md: FETCH_SIZE under-reports wide (16 B/lane) coalesced streams by 2x on gfx950 and is
# "uncalibrated for other widths: calibrate on a known byte count in your own access pattern".  These kernels read
# 4 and 12 bytes per lane; calibration on the fused kernel of config 2 (compulsory G-buffer read 48 B x 2 073 600 px
# = 99.5 MB, raw FETCH_SIZE 113 MB including the scene) gives a factor of 1.0, which is the default here.
FETCH_FACTOR = args.fetch_factor
os.makedirs(args.dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(args.dir, "c%d_%s" % (args.config, ctr))
    cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--config", str(args.config)] + (["--spp", str(args.spp)] if args.spp else [])
    subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=600)
    per_kernel = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "hrt_" not in k or "true>" in k or "math_probe" in k:
                continue
            per_kernel[k.split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    res[ctr] = per_kernel
frames = max(1, len([v for k, vs in res["FETCH_SIZE"].items() if "primary" in k for v in vs]))
out = {"config": args.config, "spp": args.spp or None, "frames": frames, "kernels": {}}
tot_path = 0.0
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    f = sum(res["FETCH_SIZE"].get(k, [])) / frames * 1024.0 * FETCH_FACTOR
    w = sum(res["WRITE_SIZE"].get(k, [])) / frames * 1024.0
    out["kernels"][k] = {"fetch_bytes_per_frame": f, "write_bytes_per_frame": w, "dispatches_per_frame": len(res["FETCH_SIZE"].get(k, [])) / frames}
    if "primary" not in k:
        tot_path += f + w
out["hbm_bytes_per_launch"] = tot_path
out["note"] = "path-trace launch = every kernel of the stage except hrt_primary_kernel; FETCH_SIZE/WRITE_SIZE in KiB; FETCH factor %.1f (calibrated, see script header)" % FETCH_FACTOR
print(json.dumps(out, indent=1))
if args.out:
    json.dump(out, open(args.out, "w"), indent=1)
