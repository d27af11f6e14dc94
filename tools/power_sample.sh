#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while bench.py runs: is the cfg2 step power-limited?
# usage (GPU box): bash tools/power_sample.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -iE "power|sclk|mclk" | head -8
python $R/bench.py --reps 40 --steps 100 --no-cpu-baseline --no-stage-sets "$@" > /tmp/ps_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "Average Graphics Package Power|Current Socket|sclk" | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python -c "import json; r=json.loads(open('/tmp/ps_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', r['ms_per_step'])"
