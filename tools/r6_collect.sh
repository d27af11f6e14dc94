#!/bin/bash
# round 6: everything profiles/r06_* holds, from one build, on one box:  bash tools/r6_collect.sh
#   bench line + bench-run kernel trace + PMC passes + stage-set traces (tools/collect_profiles.sh), the serial-engine trace (true kernel
#   durations), the registration alone, the hard route on a serial engine, the att_chain phase clocks (dev build)
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles_r06; mkdir -p $OUT
bash tools/collect_profiles.sh $OUT > $OUT/collect.log 2>&1
cd /tmp
D=/tmp/prof_serial; rm -rf $D
ENG_SERIAL=1 rocprofv3 --kernel-trace --stats -d $D -o ser -- python $R/tools/engine_timeline.py 24 > /tmp/ser.log 2>&1
python $R/tools/rocpd_summary.py $D/ser_results.db --between "match_mx6_screen_w4" > $OUT/serial_kernel_stats.md
D=/tmp/prof_reg; rm -rf $D
python $R/tools/time_pointdsc_batch.py 64 2>&1 | tail -1 > $OUT/reg_alone_time.txt
rocprofv3 --kernel-trace --stats -d $D -o reg -- python $R/tools/time_pointdsc_batch.py 64 > /tmp/reg.log 2>&1
python $R/tools/rocpd_summary.py $D/reg_results.db > $OUT/reg_alone_kernel_stats.md
D=/tmp/prof_hard; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o hards -- env ENG_HARD=1 ENG_SERIAL=1 python $R/tools/engine_timeline.py 20 > /tmp/hard.log 2>&1
python $R/tools/rocpd_summary.py $(find $D -name "hards*.db" | head -1) --between gather_mx6_v4 > $OUT/hard_serial_kernel_stats.md 2>&1
cd $R
ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step" > $OUT/hard_pipelined_step.txt
bash tools/r6_reg_clocks.sh _final > /dev/null 2>&1; cp gpurun_out/r6_reg_clocks_final.log $OUT/att_chain_phase_clocks.txt
ls -la $OUT
