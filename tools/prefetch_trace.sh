#!/bin/bash
# Per-kernel durations of the cfg2 step, plain (-1) and pipelined with knob 17 = 1 / 2 / 3 / 0 (where the next batch's gather rides), and -- when
# build/libdmvae_hip_abl12.so exists -- with gather riders that return at once: rocprofv3 --kernel-trace --stats, the program straight after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name, knob
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -o $1 -- python3 tools/pf_trace.py $2 > /dev/null 2> gpurun_out/pf_trace_$1.err
  cp $(ls gpurun_out/prof_$1/*/*kernel_stats.csv gpurun_out/prof_$1/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/pf_$1_kernel_stats.csv
  python3 - $1 <<'PY'
import csv, sys
tot = 0.0; rows = []
for r in csv.DictReader(open("gpurun_out/pf_%s_kernel_stats.csv" % sys.argv[1])):
    n = int(r["Calls"])
    if "dmvae" not in r["Name"] or n < 290: continue
    tot += float(r["TotalDurationNs"]) / 302.0 / 1e3
    if "riders" in r["Name"] or "gather" in r["Name"] or "<64, 64, 1, 4" in r["Name"]: rows.append("%s x%d %.2f us" % (r["Name"].split("(")[0].replace("void dmvae::", ""), n, float(r["AverageNs"]) / 1e3))
print("%-14s kernels per step %.1f us | %s" % (sys.argv[1], tot, " | ".join(rows)), flush=True)
PY
}
run plain -1; run k17_1 1; run k17_2 2; run k17_3 3; run k17_0 0
if [ -f deep-mixture-vae_amd/build/libdmvae_hip_abl12.so ]; then export DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl12.so; run empty_k17_1 1; run empty_k17_3 3; run empty_k17_2 2; fi
