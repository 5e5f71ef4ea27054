#!/bin/bash
# rocprofv3 kernel stats of tools/ab_bench.py for the given configs (one profile per config).  Usage: tools/prof_configs.sh "4 5" tag
set -o pipefail
export TMPDIR=/tmp
TAG=${2:-prof}
for c in $1; do
  OUT=gpurun_out/${TAG}_c$c
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/ab_bench.py --configs $c --modes auto --frames 5 > $OUT.log 2>&1 || exit 1
  grep cfg $OUT.log
  python tools/kstats.py $OUT | head -14
done
