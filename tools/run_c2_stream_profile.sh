export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2s -- python tools/ab_bench.py --configs 2 --frames 5 > gpurun_out/prof_c2s.log 2>&1
python tools/kstats.py gpurun_out/prof_c2s | grep -v "true>" | head -12
