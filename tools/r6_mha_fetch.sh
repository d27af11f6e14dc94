#!/bin/bash
# round 6: L2 hit / miss counts of the towers' attention (mha_x3_kernel) stand-alone
# -> gpurun_out/r6_mha_fetch.md      (bash tools/r6_mha_fetch.sh on the GPU box)
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
{
  echo "# rocprofv3 PMC passes: tools/r6_mha_min.py (128 images x 577 tokens x 16 heads), kernel /mha_x3/"
  # (a FETCH_SIZE / WRITE_SIZE pass over this driver aborted inside rocprofv3 on the box it was tried on: left out)
  for SET in "TCC_HIT_sum TCC_MISS_sum"; do
    P=/tmp/pmc_mha; rm -rf $P
    timeout 150 rocprofv3 --pmc $SET --kernel-trace --kernel-include-regex "mha_x3" -d $P -o p -- python $R/tools/r6_mha_min.py > /tmp/pmc_mha.log 2>&1 || tail -3 /tmp/pmc_mha.log
    echo; echo "## $SET"; python $R/tools/rocpd_summary.py $P/p_results.db | sed -n '/## PMC counters/,$p' | tail -n +3
  done
} > $R/gpurun_out/r6_mha_fetch.md 2>&1
cat $R/gpurun_out/r6_mha_fetch.md | cut -c1-200
