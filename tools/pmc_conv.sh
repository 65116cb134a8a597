#!/bin/bash
# PMC counters of the bf16 conv kernels (full pipeline probe, fast mode), separate passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 tools/full_probe.py 64 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 tools/full_probe.py 64 > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/pmc_conv/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_bf16_kernel" in k:
            key = k[k.find("ConvBfCfg"):k.find(">")+1]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
for key in sorted(agg):
    a = {c: agg[key][c] / n[key][c] for c in agg[key]}
    print(key, {c: round(v) for c, v in sorted(a.items())})
    if "GRBM_GUI_ACTIVE" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        print("   mfma busy", round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 3), " valu/mfma", round((a["SQ_INSTS_VALU"] - a["SQ_INSTS_MFMA"]) / a["SQ_INSTS_MFMA"], 2),
              " wait_inst_any/wave_cycles", round(a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 3), " lds bank conflict/idx active", round(a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
              " wait_inst_lds/wave_cycles", round(a.get("SQ_WAIT_INST_LDS", 0) / a["SQ_WAVE_CYCLES"], 3))
PY
