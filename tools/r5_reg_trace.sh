#!/bin/bash
# round 5: stand-alone kernel trace of 64 registrations (no other stream busy): per-kernel durations without pipeline contention
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
python $R/tools/time_pointdsc_batch.py 64 2>&1 | tail -1
D=/tmp/prof_reg; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o reg -- python $R/tools/time_pointdsc_batch.py 64 > /tmp/reg.log 2>&1
python $R/tools/rocpd_summary.py $D/reg_results.db > $R/gpurun_out/r6_reg_alone_kernel_stats${1}.md
cat $R/gpurun_out/r6_reg_alone_kernel_stats${1}.md | cut -c1-160
