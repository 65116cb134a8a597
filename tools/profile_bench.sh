#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes over bench.py, outputs under gpurun_out/prof/ and gpurun_out/prof_all/.
# HEADLINE command (bench.py's timed workload alone: fp32 lifting head, B = 256, no secondary legs):
#   pass 1: kernel trace + stats; pass 2/3: HBM read / write PMC counters (separate passes, no tracing mixed in);
#   pass 4: MFMA busy / clock counters.
# ALL-LEGS command (every secondary leg too): kernel trace + stats only.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof
ALL=gpurun_out/prof_all
rm -rf $OUT $ALL; mkdir -p $OUT $ALL
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --lift-only --no-fast-mode --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS --no-kernel-timing > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS --no-kernel-timing > $OUT/pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS --no-kernel-timing > $OUT/pmc_mfma.log 2>&1
echo "mfma done"
FULL="bench.py --steps 2 --warmup 1 --full-steps 1 --train-steps 1 --train-batch-bf16 256 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $ALL/trace -- python3 $FULL > $ALL/trace.log 2>&1
echo "all-legs trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $ALL/pmc_mfma -- python3 $FULL --no-kernel-timing > $ALL/pmc_mfma.log 2>&1
echo "all-legs mfma done"
du -sh $OUT $ALL
