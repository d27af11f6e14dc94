#!/bin/bash
# round 5: kernel durations of the cfg2 step WITHOUT co-runners: the serial engine (one stream) under rocprofv3
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
D=/tmp/prof_serial; rm -rf $D
ENG_SERIAL=1 rocprofv3 --kernel-trace --stats -d $D -o ser -- python $R/tools/engine_timeline.py 24 > /tmp/ser.log 2>&1
python $R/tools/rocpd_summary.py $D/ser_results.db --between "match_mx6_screen_w4" > $R/gpurun_out/r5_serial_kernel_stats.md
head -30 $R/gpurun_out/r5_serial_kernel_stats.md | cut -c1-150
