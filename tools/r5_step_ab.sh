#!/bin/bash
# round 5: pipelined cfg2 step with the registration's new kernels on / off (development library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for rep in 1 2; do
echo "all new";            python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
echo "old seeds";          ORYON_PDSC_FUSED_SEEDS=0 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
echo "old hyp";            ORYON_PDSC_FUSED_HYP=0 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
echo "all old (+K0v3)";    ORYON_K0V4=0 ORYON_PDSC_FUSED_SEEDS=0 ORYON_PDSC_FUSED_HYP=0 python tools/engine_timeline.py 40 2>&1 | grep "ms/step"
done
python tools/engine_timeline.py 40 2>&1 | tail -14
} 2>&1 | tee gpurun_out/r5_step_ab.log
