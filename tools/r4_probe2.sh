#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== correctness"
python -m pytest tests/test_gpu_matcher.py tests/test_gpu_default_route_vs_oracle.py -x -q -m gpu 2>&1 | tail -4
echo "== x3 debug stats"
ORYON_X3_DEBUG=1 ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 2 2>&1 | grep "x3" | tail -4
echo "== hard step pipelined"
ENG_HARD=1 python tools/engine_timeline.py 30 2>&1 | grep "ms/step"
echo "== hard step serial sections"
ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 10 2>&1 | tail -2
echo "== headline"
python tools/engine_timeline.py 30 2>&1 | grep "ms/step"
echo "== C=32 @192 serial"
ENG_SERIAL=1 ENG_H=192 ENG_C=32 python tools/engine_timeline.py 10 2>&1 | tail -2
} > gpurun_out/r4_probe2.log 2>&1
cat gpurun_out/r4_probe2.log
