#!/bin/bash
# SQ counters of the fused path-trace kernel on config 2 (separate passes).  Usage: tools/pmc_c2.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_c2}
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/p$i.json 2> $OUT/p$i.err
done
python tools/pmc_summary2.py $OUT | grep -A1 "path_trace"
