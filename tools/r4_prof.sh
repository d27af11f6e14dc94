#!/bin/bash
# round-4 probe: rocprofv3 kernel trace of a few steps of tools/engine_timeline.py.  usage: r4_prof.sh <tag> [env assignments...]
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/prof_$tag; mkdir -p $out
D=/tmp/prof_$tag; rm -rf $D; cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d $D -o t -- python $R/tools/engine_timeline.py 6 > $out/run.log 2>&1
python $R/tools/rocpd_summary.py $D/t_results.db > $out/kernel_stats.md
head -36 $out/kernel_stats.md | cut -c1-190
grep "ms/step" $out/run.log
