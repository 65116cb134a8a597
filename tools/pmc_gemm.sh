#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 tools/gemm_one.py ${TILE:-12} > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/p2 -- python3 tools/gemm_one.py ${TILE:-12} > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $OUT/p3 -- python3 tools/gemm_one.py ${TILE:-12} > $OUT/p3.log 2>&1
tail -2 $OUT/p3.log
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_gemm/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ("gemm_f32_persist" in r["Kernel_Name"] or "gemm_bf16_persist" in r["Kernel_Name"]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print(k, agg[k] / n[k])
PY
