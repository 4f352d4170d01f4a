#!/bin/bash
# Per-kernel SQ counters of the whole step (eager, 20 steps): MFMA busy share, LDS bank conflicts,
# wait / active cycles.  Two --pmc passes, kernel trace only (gpurun refuses --pmc with sys/hip traces).
# MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x clock):
# the counter sums the cycles every SIMD's matrix core is busy (MI355X_MICROARCH.md: = 16 per
# 16x16x32 bf16 MFMA).  Writes gpurun_out/<tag>_pmc_sq.txt :   tools/pmc_sq.sh r01     (CFG=cfg5: that config -> <tag>_cfg5_pmc_sq.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
CFG=${CFG:-}
TAG=$TAG${CFG:+_$CFG}
CMD="python3 bench.py ${CFG:+--config $CFG} --steps ${PMC_STEPS:-20} --warmup 5 --repeats 1 --no-cpu-baseline --elbo-epochs 0 --profile-steps 0 --no-graph"
rm -rf gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_sq/a -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/pmc_sq/b -- $CMD > /dev/null 2>&1
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
lines = ["# per kernel, averages over its launches; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz nominal): the chip holds",
         "# 1.9-2.1 GHz under bf16 MFMA load (tools/clock256.py), so the share of the HELD clock is ~1.2x this figure; lds_share = SQ_LDS_IDX_ACTIVE /",
         "# (256 CUs x duration x 2.4 GHz); wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls), wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier)",
         "%-52s %9s %13s %9s %11s %12s %9s %9s %9s %12s" % ("kernel", "avg_us", "mfma_busy", "mfma_util", "insts_mfma", "lds_conflict", "lds_share", "wait_inst", "wait_any", "mfma_coexec")]
for k in sorted(agg, key=lambda k: -sum(dur[k])):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    us = sum(dur[k]) / len(dur[k]) / 1e3
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = busy / (4 * 256 * us * 1e-6 * 2.4e9) if us > 0 else 0.0      # nominal 2.4 GHz
    ldss = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256 * us * 1e-6 * 2.4e9) if us > 0 else 0.0
    wc = max(1.0, c.get("SQ_WAVE_CYCLES", 1.0))
    lines.append("%-52s %9.2f %13.0f %8.1f%% %11.0f %12.0f %8.1f%% %8.1f%% %8.1f%% %12.0f" % (
        k[:52], us, busy, 100 * util, c.get("SQ_INSTS_MFMA", 0), c.get("SQ_LDS_BANK_CONFLICT", 0), 100 * ldss,
        100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 100 * c.get("SQ_WAIT_ANY", 0.0) / wc, c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0)))
open("gpurun_out/%s_pmc_sq.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:20]))
PY
