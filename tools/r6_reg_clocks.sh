#!/bin/bash
# round 6: where the 64 us of one pdsc_att_chain_x3_kernel launch go - phase clocks of two waves of workgroup 0 (dev build, ORYON_PDSC_CLOCKS = layer),
# and the registration alone with the attention / chain as separate launches for comparison
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== 64 registrations alone (shipped schedule)"
python tools/r5_time_reg.py
echo "== phase clocks of layer 5 (us since kernel entry; marks: 1 q split, 2 tiles done, 3 merged, 4 weights landed, 5 W1, 6 W2, 7 W3, 8 PointCN, 9 q chunk ready, 10 k chunk ready, 11 v chunk ready, 12 end)"
ORYON_PDSC_CLOCKS=5 python tools/r5_time_reg.py 2>&1 | grep -m 6 "att_chain clocks"
echo "== attention as its own launch (ORYON_PDSC_FUSED_ATT=0)"
ORYON_PDSC_FUSED_ATT=0 python tools/r5_time_reg.py
} 2>&1 | tee gpurun_out/r6_reg_clocks${1}.log
