#!/bin/bash
# Per-kernel SQ counters of the whole step (eager, 20 steps): MFMA busy share, LDS bank conflicts,
# wait / active cycles.  Two --pmc passes, kernel trace only (gpurun refuses --pmc with sys/hip traces).
# MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x clock):
# the counter sums the cycles every SIMD's matrix core is busy (MI355X_MICROARCH.md: = 16 per
# 16x16x32 bf16 MFMA).  Writes gpurun_out/<tag>_pmc_sq.txt :   tools/pmc_sq.sh r01
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --no-graph"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_sq/a -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq/b -- $CMD > /dev/null 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, collections, re, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for sub in "ab":
    for f in glob.glob("gpurun_out/pmc_sq/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "dmvae::" not in r["Kernel_Name"]: continue
            m = re.search(r"dmvae::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("gpurun_out/pmc_sq/%s/*/*kernel_trace.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "dmvae::" not in r["Kernel_Name"]: continue
            m = re.search(r"dmvae::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            dur[m.group(1)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
lines = ["%-46s %9s %12s %10s %12s %12s %12s %10s" % ("kernel", "avg_us", "mfma_busy", "mfma_util", "insts_mfma", "lds_conflict", "lds_active", "wait/wave")]
for k in sorted(agg, key=lambda k: -sum(dur[k])):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    us = sum(dur[k]) / len(dur[k]) / 1e3
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = busy / (4 * 256 * us * 1e-6 * 2.4e9) if us > 0 else 0.0      # nominal 2.4 GHz
    wait = c.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1.0))
    lines.append("%-46s %9.2f %12.0f %9.1f%% %12.0f %12.0f %12.0f %9.1f%%" % (k[:46], us, busy, 100 * util, c.get("SQ_INSTS_MFMA", 0), c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0), 100 * wait))
open("gpurun_out/%s_pmc_sq.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
PY
