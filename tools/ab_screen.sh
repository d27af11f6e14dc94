#!/bin/bash
# A/B of screen-kernel builds on ONE box (clocks differ between boxes): serial screen time and pipelined step for each setting.
# usage: bash tools/ab_screen.sh    (run on the GPU box; rebuilds screen_mx6.o in place)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  echo "== $1"
  for v in ${VARS:-0 2}; do
    s=$(ORYON_MX6_VAR=$v ENG_SERIAL=1 python $R/tools/engine_timeline.py 30 2>&1 | tail -1)
    b=$(ORYON_MX6_VAR=$v python $R/bench.py --reps 3 --no-cpu-baseline --no-stage-sets 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), r['timing']['stream_busy_ms_per_step'])")
    echo "VAR $v serial: $s"
    echo "VAR $v pipelined: $b"
  done
}
run "as built"
if [ -n "$AB_NNAN" ]; then
sed -i 's/^screen_mx6.o: CXXFLAGS += -fno-honor-nans/screen_mx6.o: CXXFLAGS += /' $R/oryon_amd/csrc/Makefile
touch $R/oryon_amd/csrc/screen_mx6.hip; make -s -C $R/oryon_amd/csrc 2>&1 | tail -2
run "without -fno-honor-nans"
fi
