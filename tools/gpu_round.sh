#!/bin/bash
# One GPU-box round: GPU tests, smoke, bench, rocprofv3 kernel stats, PMC traffic.  Usage: tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "host: nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" | tee $OUT/progress.log
echo "== pytest -m gpu" | tee -a $OUT/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/progress.log
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
echo "== smoke" | tee -a $OUT/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/progress.log
tail -2 $OUT/smoke.log | tee -a $OUT/progress.log
for c in 2 3; do
  echo "== bench config $c" | tee -a $OUT/progress.log
  timeout -k 10 600 python bench.py --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; echo "bench rc=$?" | tee -a $OUT/progress.log
  cat $OUT/bench_c$c.json | tee -a $OUT/progress.log
  echo "== rocprofv3 kernel stats config $c" | tee -a $OUT/progress.log
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c$c -- python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/prof_bench_c$c.json 2> $OUT/prof_c$c.err; echo "rocprof rc=$?" | tee -a $OUT/progress.log
  python tools/kstats.py $OUT/prof_c$c | head -12 | tee -a $OUT/progress.log
  echo "== PMC traffic config $c" | tee -a $OUT/progress.log
  timeout -k 10 900 python tools/collect_traffic.py --config $c --dir $OUT/traffic --out $OUT/traffic_config$c.json > $OUT/traffic_c$c.log 2>&1; echo "traffic rc=$?" | tee -a $OUT/progress.log
  grep hbm_bytes_per_launch $OUT/traffic_config$c.json | tee -a $OUT/progress.log
done
echo "== bench + kernel stats config 4 (4 spp of 64)" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py --config 4 --spp 4 --cpu-seconds 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?" | tee -a $OUT/progress.log
cat $OUT/bench_c4.json | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -- python bench.py --config 4 --spp 4 --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/prof_bench_c4.json 2> $OUT/prof_c4.err; echo "rocprof rc=$?" | tee -a $OUT/progress.log
python tools/kstats.py $OUT/prof_c4 | head -12 | tee -a $OUT/progress.log
echo "== all configs, kernel organisations" | tee -a $OUT/progress.log
timeout -k 10 600 python tools/ab_bench.py --configs 2,3,4,5 --frames 5 2>&1 | tee -a $OUT/progress.log
echo "== scene updates on the device (TLAS refit / rebuild, sphere and vertex updates)" | tee -a $OUT/progress.log
timeout -k 10 300 python tools/bvh_update_bench.py --counts 10000,100000 --out $OUT/bvh_update_bench.json > $OUT/bvh_update_bench.log 2>&1; echo "bvh_update_bench rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 python tools/mesh_update_bench.py --out $OUT/mesh_update_bench.json > $OUT/mesh_update_bench.log 2>&1; echo "mesh_update_bench rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 python bench.py --config 3 --device-tlas --cpu-seconds 0 > $OUT/bench_c3_device_tlas.json 2> $OUT/bench_c3_device_tlas.err; echo "bench --device-tlas rc=$?" | tee -a $OUT/progress.log
cut -c1-400 $OUT/bench_c3_device_tlas.json | tee -a $OUT/progress.log
STATS=ilgpu_raytracing_amd/csrc/variants/libhip_raytrace_stats.so
if [ -e $STATS ]; then
  echo "== walker phase statistics (variant build -DHRT_WALK_STATS)" | tee -a $OUT/progress.log
  HRT_LIB=$PWD/$STATS timeout -k 10 300 python tools/walk_stats.py --configs 3,4,5 2>&1 | tee $OUT/walker_phase_stats.txt | tee -a $OUT/progress.log
fi
