#!/bin/bash
# A/B of an environment switch on ONE box: bench.py ms_per_step (and the hard-descriptor step) for each value.
# usage: bash tools/ab_env.sh VAR "v1 v2 ..." [hard]
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2
for rep in 1 2; do
for v in $VALS; do
  b=$(env $VAR=$v python $R/bench.py --reps 5 --no-cpu-baseline --no-stage-sets 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3))")
  echo "$VAR=$v: $b ms per step"
  if [ -n "$3" ]; then
    h=$(env $VAR=$v ENG_HARD=1 python $R/tools/engine_timeline.py 30 2>&1 | tail -2 | awk '{print $4}' | tr -d 'M[' | paste -sd' ' | awk '{printf "%.2f", $2-$1}')
    echo "$VAR=$v: hard step $h ms"
  fi
done
done
