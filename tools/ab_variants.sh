#!/bin/bash
# Times every A/B build under ilgpu_raytracing_amd/csrc/variants/ (and the default library) on the same configs.
#   tools/ab_variants.sh "3,4,5" [frames]
set -e
cd "$(dirname "$0")/.."
CFG=${1:-3,4,5}; FR=${2:-5}
mkdir -p gpurun_out
echo "== default" | tee -a gpurun_out/ab_variants.log
python tools/ab_bench.py --configs $CFG --frames $FR --modes auto 2>&1 | tee -a gpurun_out/ab_variants.log
for so in ilgpu_raytracing_amd/csrc/variants/*.so; do
  [ -e "$so" ] || continue
  echo "== $so" | tee -a gpurun_out/ab_variants.log
  HRT_LIB=$PWD/$so python tools/ab_bench.py --configs $CFG --frames $FR --modes auto 2>&1 | tee -a gpurun_out/ab_variants.log
done
