#!/bin/bash
cd "$(dirname "$0")/.."
{
echo "== tests"
python -m pytest tests/test_gpu_native_engine.py tests/test_gpu_default_route_vs_oracle.py tests/test_gpu_matcher.py -x -q -m gpu 2>&1 | tail -4
echo "== hard step pipelined: prefetch on / off / on / off"
ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
ENG_X3_PREFETCH=0 ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
ENG_X3_PREFETCH=0 ENG_HARD=1 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
echo "== hard serial sections (prefetch on)"
ENG_SERIAL=1 ENG_HARD=1 python tools/engine_timeline.py 12 2>&1 | tail -2
echo "== headline"
python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r4_probe4.log
