#!/bin/bash
# A/B of environment settings of the tuning build: tools/ab_env2.sh "<configs>" <steps> "VAR=a VAR2=b" "VAR=c" ...   (each quoted group is one setting)
CFGS=$1; STEPS=$2; shift 2
export HRT_LIB=ilgpu_raytracing_amd/csrc/variants/libhip_raytrace_tuning.so
for c in $CFGS; do for setting in "$@"; do
  env $setting timeout -k 10 300 python bench.py --config $c --steps $STEPS --warmup 2 --cpu-seconds 0 --pmc off --extras 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j=json.loads(l); print('config $c  %-36s ms/step %.3f' % ('$setting', j['ms_per_step']))"
done; done
