#!/bin/bash
# round-4 probe: hard step with / without the seeded K1x3 scan, narrow-descriptor step (C = 32 @ 192^2), serial section times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== correctness: lazy-route tests + default route vs oracle"
python -m pytest tests/test_gpu_matcher.py tests/test_gpu_default_route_vs_oracle.py tests/test_gpu_native_engine.py -x -q -m gpu 2>&1 | tail -5
echo "== hard step, seed on"
ENG_HARD=1 python tools/engine_timeline.py 30 2>&1 | head -1
echo "== hard step, seed off"
ORYON_X3_SEED=0 ENG_HARD=1 python tools/engine_timeline.py 30 2>&1 | head -1
echo "== hard step serial, seed on / off (sections)"
ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 10 2>&1 | tail -3
ORYON_X3_SEED=0 ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 10 2>&1 | tail -3
echo "== x3 debug stats (seed on / off)"
ORYON_X3_DEBUG=1 ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 2 2>&1 | grep "x3" | tail -2
ORYON_X3_SEED=0 ORYON_X3_DEBUG=1 ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 2 2>&1 | grep "x3" | tail -2
echo "== headline step"
python tools/engine_timeline.py 30 2>&1 | head -1
echo "== C=32 @192 pipelined / serial"
ENG_H=192 ENG_C=32 python tools/engine_timeline.py 30 2>&1 | head -1
ENG_SERIAL=1 ENG_H=192 ENG_C=32 python tools/engine_timeline.py 10 2>&1 | tail -3
} > gpurun_out/r4_probe1.log 2>&1
cat gpurun_out/r4_probe1.log
