#!/bin/bash
# What each stage costs INSIDE the pipelined cfg2 step of the native engine: the step with K0's gathers (1), the matcher (2), the registration (4)
# left out after the buffers are warm (ORYON_ENGINE_ABLATE, engine.hip).  usage (GPU box): bash tools/ablate_native.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 1 4 5; do
  v=$(ORYON_ENGINE_ABLATE=$m python $R/bench.py --reps 3 --no-cpu-baseline --no-stage-sets 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3))")
  echo "ablate mask $m (1 = no K0 gathers, 2 = no matcher, 4 = no registration): $v ms per step"
done
