#!/bin/bash
# One GPU-box round: GPU tests, smoke, bench, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "host: nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) quota_v1=$(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)" | tee $OUT/progress.log
echo "== pytest -m gpu" | tee -a $OUT/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/progress.log
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.log
echo "== smoke" | tee -a $OUT/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/progress.log
tail -2 $OUT/smoke.log | tee -a $OUT/progress.log
echo "== bench" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/progress.log
cat $OUT/bench.json | tee -a $OUT/progress.log
echo "== rocprofv3 kernel stats" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" | tee -a $OUT/progress.log
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat | head -12 | tee -a $OUT/progress.log
