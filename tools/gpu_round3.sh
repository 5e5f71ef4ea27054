#!/bin/bash
# Round-3 GPU-box pass.  Usage: tools/gpu_round3.sh <tag> [tests|bench|tiles|rehearsal|all]
set -o pipefail
TAG=${1:-r03a}; WHAT=${2:-all}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=6 > $OUT/pytest_gpu.log 2>&1; RC=$?; echo "pytest rc=$RC"; tail -12 $OUT/pytest_gpu.log
  [ $RC -ne 0 ] && exit $RC
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  SECONDS=0
  timeout -k 10 1100 python bench.py --save-pmc $OUT/pmc_config2.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? in ${SECONDS}s"
  python - <<PY
import json
j=json.load(open("$OUT/bench.json"))
r=j["roofline"]; print("config2 value", j["value"], "ms", j["ms_per_step"], "frac", r["frac"], "useful", r.get("useful_lane_frac"), "issue", r.get("frac_issue_cycles"), "src", r["pmc_source"]["kind"])
for k,v in j["extra"].items():
    if isinstance(v,dict) and "mrays_per_s" in v:
        rr=v.get("roofline",{}); print(k, v["mrays_per_s"], "Mrays/s", v["ms_per_step"], "ms  frac", rr.get("frac"), "useful", rr.get("useful_lane_frac"), "issue", rr.get("frac_issue_cycles"), "traffic GB", (rr.get("traffic") or 0)/1e9, "x compulsory", (rr.get("hbm") or {}).get("traffic_over_compulsory"), "pmc s", rr.get("pmc_source",{}).get("seconds"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
PY
  tail -3 $OUT/bench.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = tiles ]; then
  timeout -k 10 300 python tools/tile_projection.py --config 3 --frames 4 --out $OUT/tile_scaling_config3.json > $OUT/tiles3.log 2>&1; echo "tiles3 rc=$?"; tail -2 $OUT/tiles3.log
  timeout -k 10 600 python tools/tile_projection.py --config 4 --frames 1 --out $OUT/tile_scaling_config4.json > $OUT/tiles4.log 2>&1; echo "tiles4 rc=$?"; tail -6 $OUT/tiles4.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = profiles ]; then
  for spec in "2:0" "3:0" "4:4"; do
    c=${spec%%:*}; spp=${spec#*:}; SPP=""; [ "$spp" != 0 ] && SPP="--spp $spp"
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c$c -- python bench.py --config $c $SPP --steps 20 --warmup 5 --cpu-seconds 0 --pmc off --extras 0 > $OUT/prof_bench_c$c.json 2> $OUT/prof_c$c.err; echo "rocprof config $c rc=$?"
    python tools/kstats.py $OUT/prof_c$c 12
    cp $(find $OUT/prof_c$c -name "*kernel_stats.csv" | head -1) $OUT/config${c}_kernel_stats.csv
    find $OUT/prof_c$c -name "*.csv" -delete; find $OUT/prof_c$c -name "*.db" -delete
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = rehearsal ]; then
  HRT_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_rehearsal2.json 2> $OUT/bench_rehearsal2.err; echo "rehearsal rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("$OUT/bench_rehearsal2.json")); r=j["roofline"]; print("rehearsal N=2", j["value"], "frac", r["frac"], "src", r["pmc_source"], "priced", r.get("priced"))
except Exception as e: print("rehearsal parse failed", e)
PY
  tail -3 $OUT/bench_rehearsal2.err
fi
