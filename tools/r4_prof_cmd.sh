#!/bin/bash
# round-4 probe: rocprofv3 kernel trace of an arbitrary command.  usage: r4_prof_cmd.sh <tag> <grep-regex> <cmd...>
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; rx=$2; shift 2
out=$R/gpurun_out/prof_$tag; mkdir -p $out
D=/tmp/prof_$tag; rm -rf $D; cd /tmp
ARGS=(); for a in "$@"; do if [ -e "$R/$a" ] && [ "${a:0:1}" != "/" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done
rocprofv3 --kernel-trace --stats -d $D -o t -- "${ARGS[@]}" > $out/run.log 2>&1
python $R/tools/rocpd_summary.py $D/t_results.db > $out/kernel_stats.md
grep -E "$rx" $out/kernel_stats.md | head -40 | cut -c1-170
tail -c 600 $out/run.log
