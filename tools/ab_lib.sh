#!/bin/bash
# A/B of variant builds (make variant NAME=...) over BASELINE configs on one box: tools/ab_lib.sh "<names>" "<configs>" [steps] [rounds]
NAMES=$1; CFGS=$2; STEPS=${3:-6}; ROUNDS=${4:-2}
for c in $CFGS; do for r in $(seq $ROUNDS); do for n in $NAMES; do
  HRT_LIB=ilgpu_raytracing_amd/csrc/variants/libhip_raytrace_$n.so timeout -k 10 300 python bench.py --config $c --steps $STEPS --warmup 2 --cpu-seconds 0 --pmc off --extras 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j=json.loads(l); print('config $c %-12s ms/step %.3f  path stage %.3f ms' % ('$n', j['ms_per_step'], j['extra']['path_trace_kernel_ms']))"
done; done; done
