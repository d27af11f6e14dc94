#!/bin/bash
cd "$(dirname "$0")/.."
for rs in 1 2 3; do for gs in 2 3 4; do
  echo -n "reg_streams=$rs gather_sets=$gs : "; ENG_REG_STREAMS=$rs ENG_GATHER_SETS=$gs python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
done; done
echo -n "att-chain off, rs=2 gs=3: "; ORYON_PDSC_FUSED_ATT=0 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
echo -n "reg_lag=1: "; ENG_REG_LAG=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
