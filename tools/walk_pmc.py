"""Cache-side PMC counters of the walk kernels (one rocprofv3 pass per counter group over `bench.py --pmc-child`):
L1 (TCP) accesses, L1 -> L2 read requests and their summed latency, L2 (TCC) requests / hits / misses, L1 stall cycles.
   python tools/walk_pmc.py --config 4 --spp 4 [--out profiles/r03_walk_cache_counters_config4.json]"""
import sys, os, json, glob, csv, shutil, subprocess, argparse, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=4); ap.add_argument("--spp", type=int, default=4); ap.add_argument("--out", default="")
a = ap.parse_args()
PASSES = [("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum"),
          ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"),
          ("TCP_GATE_EN1_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum"),
          ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU")]
per = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for i, ctrs in enumerate(PASSES):
    d = os.path.join(ROOT, "gpurun_out", "walk_pmc", "pass%d" % i); shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", str(a.config), "--spp", str(a.spp)]
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "hrt_" not in k: continue
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if i == 0 and row["Counter_Name"] == ctrs[0]: disp[k] += 1
    shutil.rmtree(d, ignore_errors=True)
frames = 2          # bench.py --pmc-child renders 1 + 1 frames of a 4K config
out = {"config": a.config, "spp": a.spp, "frames": frames, "kernels": {}}
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    v = {n: x / frames for n, x in c.items()}
    rd = v.get("TCP_TCC_READ_REQ_sum", 0.0)
    v["dispatches_per_frame"] = disp[k] / frames
    v["l1_hit_rate"] = 1.0 - rd / max(1.0, v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0))
    v["avg_l1_to_l2_read_latency_cycles"] = v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / max(1.0, rd)
    v["l2_hit_rate"] = v.get("TCC_HIT_sum", 0.0) / max(1.0, v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0))
    v["l1_to_l2_read_requests_per_gpu_cycle"] = rd / max(1.0, v.get("GRBM_GUI_ACTIVE", 0.0))
    out["kernels"][k] = v
    print("%-58s x%-4.1f L1 acc %8.1fM  L1->L2 rd %8.1fM (L1 hit %.3f)  avg L2-read latency %6.0f cyc  L2 hit %.3f  L2 rd req / GPU cycle %.2f  wait %.2f"
          % (k[:58], v["dispatches_per_frame"], v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 1e6, rd / 1e6, v["l1_hit_rate"], v["avg_l1_to_l2_read_latency_cycles"], v["l2_hit_rate"],
             v["l1_to_l2_read_requests_per_gpu_cycle"], v.get("SQ_WAIT_ANY", 0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1))), flush=True)
if a.out: json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
