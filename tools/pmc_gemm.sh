#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <layout M N K bm bn>   -- two PMC passes on one GEMM shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$tag/a -- python3 tools/one_gemm.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_$tag/b -- python3 tools/one_gemm.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "ab":
    for f in glob.glob("gpurun_out/pmc_$tag/%s/*/*counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-28s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
