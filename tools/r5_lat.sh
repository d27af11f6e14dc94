#!/bin/bash
cd "$(dirname "$0")/.."
for gs in 3 2 3 2; do echo "hard gather_sets $gs"; ENG_HARD=1 ENG_GATHER_SETS=$gs python tools/engine_timeline.py 40 2>&1 | grep "ms/step"; done
for gs in 3 2; do echo "easy gather_sets $gs"; ENG_GATHER_SETS=$gs python tools/engine_timeline.py 40 2>&1 | grep "ms/step"; done
python -m pytest tests/test_gpu_native_engine.py tests/test_gpu_pipeline.py tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -3
