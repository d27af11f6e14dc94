#!/bin/bash
# round 5: registration A/B (development switches), stand-alone trace, PointDSC tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_pointdsc.py tests/test_gpu_lift_kabsch.py -x -q -m gpu 2>&1 | tail -4
echo "== 64 registrations alone: new / old(ORYON_PDSC_FUSED_HYP=0 etc.)"
python tools/r5_time_reg.py
ORYON_PDSC_FUSED_HYP=0 ORYON_PDSC_FUSED_TAIL=0 ORYON_PDSC_FUSED_SEEDS=0 python tools/r5_time_reg.py
python tools/r5_time_reg.py
bash tools/r5_reg_trace.sh _new | tail -25
} 2>&1 | tee gpurun_out/r5_reg.log
