#!/bin/bash
# round 5: kernel trace of the hard-descriptor route (smooth fields: x3_prefetch on) + its step time
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/hard
ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep -E "ms/step|stage|K0|match|reg" | head -12
rocprofv3 --kernel-trace --stats -d gpurun_out/hard -o hard -- env ENG_HARD=1 python tools/engine_timeline.py 30 > gpurun_out/hard/run.log 2>&1
db=$(find gpurun_out/hard -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" --between match_mx6_screen_w4 > gpurun_out/hard/hard_kernel_stats.md 2>&1
head -40 gpurun_out/hard/hard_kernel_stats.md | cut -c1-200
