#!/bin/bash
# TN weight-gradient GEMM alone: duration and L2->fabric read bytes per shape and batch (usage: tools/tn_prof.sh B)
export TMPDIR=/tmp
B=${1:-1024}
mkdir -p gpurun_out/prof_tn
for s in "1024 1024" "1024 4096" "4096 1024"; do
  tag=${B}_$(echo $s | tr ' ' x)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tn/t_$tag -- python3 tools/bf16s_one.py tn $B $s > gpurun_out/prof_tn/t_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_tn/f_$tag -- python3 tools/bf16s_one.py tn $B $s > gpurun_out/prof_tn/f_$tag.log 2>&1
done
