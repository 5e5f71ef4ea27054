#!/bin/bash
# A/B of one environment knob of the tuning build over BASELINE configs: tools/ab_env.sh <VAR> "<values>" "<configs>" [steps]
# e.g. tools/ab_env.sh HRT_FUSE "0 1" "3 4 5"
VAR=$1; VALS=$2; CFGS=$3; STEPS=${4:-6}
export HRT_LIB=ilgpu_raytracing_amd/csrc/variants/libhip_raytrace_tuning.so
for c in $CFGS; do for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --config $c --steps $STEPS --warmup 2 --cpu-seconds 0 --pmc off --extras 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j=json.loads(l); print('config $c $VAR=$v  ms/step %.3f  path stage %.3f ms  %.0f Mrays/s' % (j['ms_per_step'], j['extra']['path_trace_kernel_ms'], j['value']))"
done; done
