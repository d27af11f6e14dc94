#!/bin/bash
# round 5: K0v4 on narrow maps (the reference's own C = 32 @ 192 x 192): tests + pipelined step, development library
cd "$(dirname "$0")/.."
{
ORYON_TEST_DEV_LIB=1 ORYON_K0V4=2 python -m pytest tests/test_gpu_default_route_vs_oracle.py tests/test_gpu_native_engine.py -x -q -m gpu 2>&1 | tail -3
for v in 2 1 2 1; do echo "C=32 H=192 K0V4=$v: $(ORYON_K0V4=$v ENG_C=32 ENG_H=192 python tools/engine_timeline.py 40 2>&1 | grep 'ms/step')"; done
for v in 2 1; do echo "C=32 H=192 hard K0V4=$v: $(ORYON_K0V4=$v ENG_HARD=1 ENG_C=32 ENG_H=192 python tools/engine_timeline.py 40 2>&1 | grep 'ms/step')"; done
for v in 2 1; do echo "C=96 H=192 K0V4=$v: $(ORYON_K0V4=$v ENG_C=96 ENG_H=192 python tools/engine_timeline.py 40 2>&1 | grep 'ms/step')"; done
} 2>&1 | tee gpurun_out/r5_narrow.log
