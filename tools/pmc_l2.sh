#!/bin/bash
# usage: tools/pmc_l2.sh <tag> <layout M N K bm bn>  -- L2 hit/miss + fabric traffic of one GEMM shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_$tag/c -- python3 tools/one_gemm.py "$@" > gpurun_out/pmc_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/pmc_$tag/d -- python3 tools/one_gemm.py "$@" >> gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in "cd":
    for f in glob.glob("gpurun_out/pmc_$tag/%s/*/*counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-28s %14.0f  (n=%d, last %.0f)" % (k, sum(v) / len(v), len(v), v[-1]))
PY
tail -3 gpurun_out/pmc_$tag.log | cut -c1-300
