#!/bin/bash
# round 5: PMC passes over the stand-alone CLIP attention kernel (tools/r5_mha_time.py): where do a wave's cycles go?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/mha
cd /tmp
i=0
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); P=/tmp/mha_pmc$i; rm -rf $P
  rocprofv3 --pmc $SET --kernel-trace --kernel-include-regex "mha_x3" -d $P -o p -- python $R/tools/r5_mha_time.py > /tmp/mha_pmc.log 2>&1 || tail -3 /tmp/mha_pmc.log
  db=$(find $P -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py "$db" | sed -n '/## PMC counters/,$p'
done > $R/gpurun_out/mha/pmc.md 2>&1
cat $R/gpurun_out/mha/pmc.md | cut -c1-200
