#!/bin/bash
# round 5: stand-alone kernel durations of the realistic narrow route (C = 32 @ 192 x 192, smooth descriptors): serial engine under rocprofv3
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
D=/tmp/prof_narrow; rm -rf $D
ENG_SERIAL=1 ENG_HARD=1 ENG_C=32 ENG_H=192 rocprofv3 --kernel-trace --stats -d $D -o nar -- python $R/tools/engine_timeline.py 24 > /tmp/nar.log 2>&1
tail -3 /tmp/nar.log
python $R/tools/rocpd_summary.py $D/nar_results.db --between "match_mx6_screen_w4" > $R/gpurun_out/r5_narrow_kernel_stats.md
head -45 $R/gpurun_out/r5_narrow_kernel_stats.md | cut -c1-150
