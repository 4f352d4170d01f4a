#!/bin/bash
# The fused  heads forward + latent  launch (csrc/heads_latent.hip, knob 19) against the two launches it replaces, at the metric's batch (cfg2):
# per-kernel durations (rocprofv3 --kernel-trace --stats of 300 replays of the plain captured step, tools/pf_trace.py -1), then the step A/B at several
# batch sizes (tools/knob_step.py cfgN[:batch=B] 19 0 1 1 0).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name, knobs
  DMVAE_KNOBS="$2" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -o $1 -- python3 tools/pf_trace.py -1 > /dev/null 2> gpurun_out/hl_trace_$1.err
  cp $(ls gpurun_out/prof_$1/*/*kernel_stats.csv gpurun_out/prof_$1/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/hl_$1_kernel_stats.csv
  python3 - $1 <<'PY'
import csv, sys
tot = 0.0; rows = []
for r in csv.DictReader(open("gpurun_out/hl_%s_kernel_stats.csv" % sys.argv[1])):
    n = int(r["Calls"])
    if "dmvae" not in r["Name"] or n < 290: continue
    tot += float(r["TotalDurationNs"]) / 302.0 / 1e3
    if "latent" in r["Name"] or "<64, 64, 0, 1" in r["Name"]: rows.append("%s x%d %.2f us" % (r["Name"].split("(")[0].replace("void dmvae::", ""), n, float(r["AverageNs"]) / 1e3))
print("%-10s kernels per step %.1f us | %s" % (sys.argv[1], tot, " | ".join(rows)), flush=True)
PY
  rm -rf gpurun_out/prof_$1
}
run two "19=0"; run fused "19=1"; run two_b "19=0"; run fused_b "19=1"
for c in cfg1 cfg2:batch=256 cfg2:batch=1024 cfg2:batch=2048 cfg2; do python3 tools/knob_step.py $c 19 0 1 1 0 2>&1 | grep -v amdgpu.ids | tail -1; done
