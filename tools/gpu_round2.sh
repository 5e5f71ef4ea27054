#!/bin/bash
# Round-2 GPU-box pass: GPU tests, smoke, bench (PMC collected in-run, saved), rocprofv3 kernel stats of the same command.
# Usage: tools/gpu_round2.sh <tag> [tests|bench|all]
set -o pipefail
TAG=${1:-r02a}
WHAT=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "host: nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" | tee $OUT/progress.log
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
  echo "== pytest -m gpu" | tee -a $OUT/progress.log
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; RC=$?; echo "pytest rc=$RC" | tee -a $OUT/progress.log
  tail -15 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
  [ $RC -ne 0 ] && exit $RC
  echo "== smoke" | tee -a $OUT/progress.log
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/progress.log
  tail -2 $OUT/smoke.log | tee -a $OUT/progress.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  echo "== bench (default command + --save-pmc)" | tee -a $OUT/progress.log
  timeout -k 10 900 python bench.py --save-pmc $OUT/pmc_config2.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/progress.log
  cat $OUT/bench.json | tee -a $OUT/progress.log
  tail -5 $OUT/bench.err | tee -a $OUT/progress.log
  echo "== rocprofv3 kernel stats of the bench command (config 2 only, no PMC child, no extras)" | tee -a $OUT/progress.log
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pmc off --extras 0 > $OUT/prof_bench_c2.json 2> $OUT/prof_c2.err; echo "rocprof rc=$?" | tee -a $OUT/progress.log
  python tools/kstats.py $OUT/prof_c2 8 | tee -a $OUT/progress.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = configs ]; then
  for spec in "3:0" "4:4"; do
    c=${spec%%:*}; spp=${spec#*:}; SPP=""; [ "$spp" != 0 ] && SPP="--spp $spp"
    echo "== bench config $c $SPP (+ --save-pmc)" | tee -a $OUT/progress.log
    timeout -k 10 600 python bench.py --config $c $SPP --steps 20 --warmup 3 --cpu-seconds 0 --save-pmc $OUT/pmc_config$c.json > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; echo "bench rc=$?" | tee -a $OUT/progress.log
    cut -c1-1500 $OUT/bench_c$c.json | tee -a $OUT/progress.log
    echo "== rocprofv3 kernel stats config $c" | tee -a $OUT/progress.log
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c$c -- python bench.py --config $c $SPP --steps 10 --warmup 2 --cpu-seconds 0 --pmc off > $OUT/prof_bench_c$c.json 2> $OUT/prof_c$c.err; echo "rocprof rc=$?" | tee -a $OUT/progress.log
    python tools/kstats.py $OUT/prof_c$c 12 | tee -a $OUT/progress.log
  done
fi
