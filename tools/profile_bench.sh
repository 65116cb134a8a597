#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes, outputs under gpurun_out/prof*/ (scratch); tools/summarize_prof.py digests them
# into profiles/rNN_*.  Every command gets FOUR separate passes (never combined: PMC + tracing together is refused on this pool):
#   1 kernel trace + stats (durations)   2 FETCH_SIZE   3 WRITE_SIZE   4 MFMA busy / clock / wave counters
# usage: tools/profile_bench.sh [headline] [config3] [hm16] [stage1] [all]     (default: headline config3)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
passes() {   # $1 = output dir, rest = python arguments
  local OUT=$1; shift
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1
  echo "$OUT trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 "$@" > $OUT/pmc_fetch.log 2>&1
  echo "$OUT fetch done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 "$@" > $OUT/pmc_write.log 2>&1
  echo "$OUT write done"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_mfma -- python3 "$@" > $OUT/pmc_mfma.log 2>&1
  echo "$OUT mfma done"
}
SETS="$@"; [ -z "$SETS" ] && SETS="headline config3"
for s in $SETS; do
  case $s in
    headline) passes gpurun_out/prof bench.py --steps 5 --warmup 2 --lift-only --no-fast-mode --no-cpu-baseline --no-kernel-timing ;;
    config3)  passes gpurun_out/prof_c3 tools/train_bf16_probe.py 1024 bf16 ;;
    hm16)     passes gpurun_out/prof_hm tools/hm_bf16_probe.py 256 64 bf16 ;;
    stage1)   passes gpurun_out/prof_st1 tools/stage1_probe.py 32 f32 ;;
    rgbdef)   passes gpurun_out/prof_rgbdef tools/from_rgb_probe.py 1024 default ;;
    all)      passes gpurun_out/prof_all bench.py --steps 2 --warmup 1 --full-steps 1 --train-steps 1 --no-cpu-baseline --no-kernel-timing ;;
  esac
done
du -sh gpurun_out/prof*
