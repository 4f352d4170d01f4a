#!/bin/bash
# Per-kernel durations of one config's captured step as rocprofv3 sees them in replay:  tools/step_trace.sh <cfgN>[:batch=B] [tag]   (DMVAE_KNOBS passes through)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$1; T=${2:-$(echo $1 | tr ':=' '__')}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_st_$T -o st -- python3 tools/step_trace.py $C > /dev/null 2> gpurun_out/step_trace_$T.err
cp $(ls gpurun_out/prof_st_$T/*/*kernel_stats.csv gpurun_out/prof_st_$T/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/step_trace_${T}_kernel_stats.csv
rm -rf gpurun_out/prof_st_$T
python3 - gpurun_out/step_trace_${T}_kernel_stats.csv "$C" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dmvae" in r["Name"] and int(r["Calls"]) >= 50]
reps = max(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows) / reps / 1e3
print("%s: %d kernels families, %.1f us of kernels per step (%d replays)" % (sys.argv[2], len(rows), tot, reps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("  %6.2f us x %.0f  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]) / reps, r["Name"].split("(")[0].replace("void dmvae::", "")[:110]))
PY
