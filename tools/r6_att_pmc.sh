#!/bin/bash
# round 6: PMC pass of the registration alone (pdsc_att_chain_x3_kernel: matrix-pipe busy cycles, clock, waits) + the two MFMA probes
# -> gpurun_out/r6_att_pmc.md, gpurun_out/r6_mfma_probes.txt       (bash tools/r6_att_pmc.sh on the GPU box)
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp; P=/tmp/pmc_att; rm -rf $P
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --kernel-include-regex "pdsc_att_chain" -d $P -o p -- python $R/tools/time_pointdsc_batch.py 64 > /tmp/pmc_att.log 2>&1
{
  echo "# rocprofv3 PMC pass: tools/time_pointdsc_batch.py 64 (64 registrations alone), kernels /pdsc_att_chain/"
  echo
  python $R/tools/rocpd_summary.py $P/p_results.db | sed -n '/## PMC counters/,$p' | tail -n +3
} > $R/gpurun_out/r6_att_pmc.md 2>&1
cd $R
{ echo "== tools/probe_mfma_power.hip"; tools/bin/probe_power; echo; echo "== tools/probe_mfma_shadow.hip"; tools/bin/probe_shadow; } > gpurun_out/r6_mfma_probes.txt 2>&1
tail -5 gpurun_out/r6_att_pmc.md
