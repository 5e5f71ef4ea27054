#!/bin/bash
# A/B of the default library against one variant, interleaved A B A B on the same box:  tools/ab_pair.sh <variant.so> "2,3,4" [frames] [rounds]
set -e
cd "$(dirname "$0")/.."
V=$1; CFG=${2:-2,3,4}; FR=${3:-20}; N=${4:-2}
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  echo "== default ($i)" | tee -a gpurun_out/ab_pair.log
  python tools/ab_bench.py --configs $CFG --frames $FR --modes auto 2>&1 | tee -a gpurun_out/ab_pair.log
  echo "== $V ($i)" | tee -a gpurun_out/ab_pair.log
  HRT_LIB=$PWD/$V python tools/ab_bench.py --configs $CFG --frames $FR --modes auto 2>&1 | tee -a gpurun_out/ab_pair.log
done
