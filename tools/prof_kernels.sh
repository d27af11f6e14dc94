#!/bin/bash
# usage: tools/prof_kernels.sh <regex> <cmd...>  -> prints per-kernel avg us for kernels matching regex (run on the GPU box)
# relative script paths in <cmd> are resolved against the repo root (rocprofv3 itself runs from /tmp)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; RX=$1; shift
D=$(mktemp -d /tmp/prof.XXXX); cd /tmp
ARGS=(); for a in "$@"; do if [ -e "$R/$a" ] && [ "${a:0:1}" != "/" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done
rocprofv3 --kernel-trace --kernel-include-regex "$RX" -d $D -o p -- "${ARGS[@]}" > $D/log 2>&1
python $R/tools/rocpd_summary.py $D/p_results.db | awk -F'|' 'NR>4 {n=$2; gsub(/\(.*/,"",n); printf "%-62s calls %s avg_us %s\n", substr(n,1,62), $3,$5}'
