export TMPDIR=/tmp
HRT_WIDE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide4 -- python tools/ab_bench.py --configs 4 --modes auto --frames 5 > gpurun_out/prof_wide4.log 2>&1
python tools/kstats.py gpurun_out/prof_wide4 | head -8
HRT_WIDE=1 timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_wide4/p1 -- python tools/ab_bench.py --configs 4 --modes auto --frames 3 > /dev/null 2>&1
HRT_WIDE=1 timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_wide4/p2 -- python tools/ab_bench.py --configs 4 --modes auto --frames 3 > /dev/null 2>&1
HRT_WIDE=1 timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_wide4/p3 -- python tools/ab_bench.py --configs 4 --modes auto --frames 3 > /dev/null 2>&1
python tools/pmc_summary2.py gpurun_out/pmc_wide4 | grep -A1 "walkw"
