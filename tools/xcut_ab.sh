#!/bin/bash
# The weight-gradient group's XCD partition (knob 20; csrc/gemm_bf16.hip grouped_launch): runs cut by streamed bytes over the whole problem sequence (1)
# against runs cut per tile-shape class by count (0).  Step A/B on one engine (tools/knob_step.py, a b b a), then FETCH_SIZE of the launch under both
# (one --pmc pass each, kernel-trace only; bench.py reads DMVAE_KNOBS).    tools/xcut_ab.sh [cfgs...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${@:-cfg2 cfg4 cfg3}; do
  python3 tools/knob_step.py $c 20 0 1 1 0 2>&1 | grep -v amdgpu.ids | tail -1
done
for c in ${@:-cfg2 cfg4 cfg3}; do
  for k in 0 1; do
    rm -rf gpurun_out/pmc_xc
    DMVAE_KNOBS="20=$k" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_xc -- python3 bench.py --config $c --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --elbo-epochs 0 --profile-steps 0 --no-graph > /dev/null 2>&1
    python3 - $c $k <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_xc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "grouped_kernel<64, 64, 2" in r["Kernel_Name"]: agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%s knob 20 = %s: %-70s fetch (x2) %8.1f MB per launch (%d launches)" % (sys.argv[1], sys.argv[2], k.split("(")[0][-60:], sum(v) / len(v) * 2048.0 / 1e6, len(v)), flush=True)
PY
  done
done
