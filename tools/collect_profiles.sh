#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ holds, on the GPU box:  bash tools/collect_profiles.sh <out_dir>
#   1. kernel-trace of `python bench.py` (default workload)                     -> <out>/kernel_stats.md + the printed JSON line
#   2. separate --pmc passes (no trace domains besides --kernel-trace), restricted to the heavy kernels -> <out>/pmc_counters.md
#   3. full (feat+match+pose) stage on the fp16x3 path: kernel trace + matrix-pipe counters -> <out>/full_stage_*.md
#   4. decode stage set (fusion + decoder fast path): kernel trace + traffic / matrix-pipe counters of the decoder kernels -> <out>/decode_stage_*.md
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/profiles}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp
python $R/bench.py --no-cpu-baseline --no-stage-sets > "$OUT/bench_line.json" 2> /tmp/bench.err
D=/tmp/prof_trace; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-sets > /tmp/trace.log 2>&1
python $R/tools/rocpd_summary.py $D/bench_results.db --after "distribution_elementwise|index_elementwise" > "$OUT/kernel_stats.md"
python $R/tools/rocpd_summary.py $D/bench_results.db --between "match_mx6_screen_w4" > "$OUT/kernel_stats_steps_only.md"
python $R/tools/rocpd_summary.py $D/bench_results.db > "$OUT/kernel_stats_whole_run.md"
RX='mx6_screen|screen_v2_kernel|gather_q8_v3|gather_mx6_v4|pdsc_att|match_decide|match_resolve|pdsc_linear|pdsc_pcn_qkv|pdsc_mlp3|pdsc_hyp|pdsc_seed|pdsc_head'
{
  echo "# rocprofv3 PMC passes: bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap (2 engine passes, B=64), kernels /$RX/"
  for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    P=/tmp/prof_pmc; rm -rf $P
    rocprofv3 --pmc $SET --kernel-trace --kernel-include-regex "$RX" -d $P -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-sets --no-overlap > /tmp/pmc.log 2>&1
    echo; echo "## pass: $SET"; echo
    python $R/tools/rocpd_summary.py $P/p_results.db | sed -n '/## PMC counters/,$p' | tail -n +3
  done
} > "$OUT/pmc_counters.md"
# FETCH_SIZE calibration for 4 B/lane loads (the NCHW gather's access width): 1 GiB read once per kernel
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_fetch $R/tools/probe_fetch_calib.hip 2>/dev/null && {
  P=/tmp/prof_cal; rm -rf $P
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o c -- /tmp/probe_fetch > /tmp/cal.log 2>&1
  { echo; echo "## FETCH_SIZE calibration: 1 GiB (1048576 KiB) read once by each kernel"; echo
    python $R/tools/rocpd_summary.py $P/c_results.db | sed -n '/## PMC counters/,$p' | tail -n +3; } >> "$OUT/pmc_counters.md"
}
# 3. the widened stage set on the fp32-grade fp16x3 path: kernel trace of one full (feat+match+pose) stage run and the matrix-pipe
#    counters of its GEMM / attention kernels
D=/tmp/prof_full; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o full -- python $R/bench.py --stages full --backbone-dtype fp16x3 --steps 2 --warmup 1 > /tmp/full.log 2>&1
python $R/tools/rocpd_summary.py $D/full_results.db --exclude "naive_conv|Im2d2Col|Col2Im2d" > "$OUT/full_stage_kernel_stats.md"   # MIOpen find-mode trial launches
{
  echo "# rocprofv3 PMC pass: bench.py --stages full --backbone-dtype fp16x3 --steps 1 --warmup 0, kernels /linear_f16x3_stream|mha_x3/"
  P=/tmp/prof_pmc_full; rm -rf $P
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --kernel-include-regex "linear_f16x3_stream|mha_x3" -d $P -o p -- python $R/bench.py --stages full --backbone-dtype fp16x3 --steps 1 --warmup 0 > /tmp/pmc_full.log 2>&1
  echo; python $R/tools/rocpd_summary.py $P/p_results.db | sed -n '/## PMC counters/,$p' | tail -n +3
} > "$OUT/full_stage_pmc.md"
# 4. the decode stage set (fusion + decoder on cached encodings, fast path: fp16x3 linears, HIP decoder, fused window attention):
#    kernel trace of the stage run and the traffic / matrix-pipe counters of the decoder's kernels on tools/r4_fastpath.py (128 images)
D=/tmp/prof_decode; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o dec -- python $R/bench.py --stages decode --backbone-dtype fp16x3 --steps 2 --warmup 1 > /tmp/decode.log 2>&1
python $R/tools/rocpd_summary.py $D/dec_results.db --exclude "naive_conv|Im2d2Col|Col2Im2d" > "$OUT/decode_stage_kernel_stats.md"
{
  echo "# rocprofv3 PMC passes: tools/r4_fastpath.py (fusion + decoder fast path, 128 images), kernels /dec_conv3x3|dec_final|dec_upconv|fusion_window_attention/"
  for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    P=/tmp/prof_pmc_dec; rm -rf $P
    rocprofv3 --pmc $SET --kernel-trace --kernel-include-regex "dec_conv3x3|dec_final|dec_upconv|fusion_window_attention" -d $P -o p -- python $R/tools/r4_fastpath.py > /tmp/pmc_dec.log 2>&1
    echo; echo "## pass: $SET"; echo
    python $R/tools/rocpd_summary.py $P/p_results.db | sed -n '/## PMC counters/,$p' | tail -n +3
  done
} > "$OUT/decode_stage_pmc.md"
ls -la "$OUT"
