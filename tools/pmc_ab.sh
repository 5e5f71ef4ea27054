#!/bin/bash
# PMC passes over tools/ab_bench.py for one config.  Usage: tools/pmc_ab.sh <tag> <config>
TAG=$1; CFG=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python tools/ab_bench.py --configs $CFG --frames 1 > $OUT/p$i.log 2> $OUT/p$i.err
  echo "pass $i rc=$?" >> $OUT/progress.log
done
python tools/pmc_summary2.py $OUT
