#!/bin/bash
# Pipelined cfg2 step for several engine geometries on ONE box (tools/engine_timeline.py reads ENG_* overrides of native_geometry).
# usage: bash tools/sweep_geometry.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" ENG_NO_TIMING= python $R/tools/engine_timeline.py 40 2>&1 | tail -12 | awk '{print $6}' | awk 'NR>1{s+=$1-p; n++} {p=$1} END{printf "%.3f", s/n}'; }
for rs in 1 2 3 4; do for ns in 6 8; do for gs in 2 3; do
  echo "reg_streams=$rs n_slots=$ns gather_sets=$gs: $(step ENG_REG_STREAMS=$rs ENG_N_SLOTS=$ns ENG_GATHER_SETS=$gs) ms per step"
done; done; done
