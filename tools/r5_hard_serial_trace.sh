#!/bin/bash
# round 5: kernel trace of the hard-descriptor route on a SERIALISED engine (true per-kernel durations: nothing runs beside a kernel)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out/hard
rocprofv3 --kernel-trace --stats -d gpurun_out/hard -o hards -- env ENG_HARD=1 ENG_SERIAL=1 python tools/engine_timeline.py 20 > gpurun_out/hard/runs.log 2>&1
db=$(find gpurun_out/hard -name "hards*.db" | head -1)
python tools/rocpd_summary.py "$db" --between gather_mx6_v4 > gpurun_out/hard/hard_serial_kernel_stats.md 2>&1
head -40 gpurun_out/hard/hard_serial_kernel_stats.md | cut -c1-175
