#!/bin/bash
cd "$(dirname "$0")/.."
{
echo "== pointdsc + pipeline tests"
python -m pytest tests/test_gpu_pointdsc.py tests/test_gpu_native_engine.py tests/test_gpu_default_route_vs_oracle.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -4
echo "== serial sections: att-chain on / off"
ENG_SERIAL=1 python tools/engine_timeline.py 10 2>&1 | tail -2
ORYON_PDSC_FUSED_ATT=0 ENG_SERIAL=1 python tools/engine_timeline.py 10 2>&1 | tail -2
echo "== pipelined, on / off / on / off"
python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
ORYON_PDSC_FUSED_ATT=0 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
ORYON_PDSC_FUSED_ATT=0 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r4_probe3.log
