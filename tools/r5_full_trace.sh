#!/bin/bash
# round 5: kernel trace of the full stage set with CLIP weights as clip.load leaves them (fp16x3-clipload)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/fullc
rocprofv3 --kernel-trace --stats -d gpurun_out/fullc -o fullc -- python bench.py --stages full --backbone-dtype ${DT:-fp16x3-clipload} --steps 5 --warmup 1 > gpurun_out/fullc/run.log 2>&1
tail -1 gpurun_out/fullc/run.log | cut -c1-300
db=$(find gpurun_out/fullc -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" --exclude "naive_conv|Im2d2Col|Col2Im2d" > gpurun_out/fullc/kernel_stats.md 2>&1
head -45 gpurun_out/fullc/kernel_stats.md | cut -c1-200
