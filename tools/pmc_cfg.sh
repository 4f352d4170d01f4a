#!/bin/bash
# one-off: FETCH_SIZE of the dW group at cfg3 / cfg4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg3 cfg4; do
  rm -rf gpurun_out/pmc_$cfg
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --elbo-epochs 0 --profile-steps 0 --no-graph > /dev/null 2>&1
  python3 - $cfg <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "dmvae::" in r["Kernel_Name"]: agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg, key=lambda k: -sum(agg[k]))[:6]:
    print(sys.argv[1], "%-70s launches %3d  fetch(x2) %9.1f MB/launch" % (k.replace("void dmvae::", "")[:70], len(agg[k]), sum(agg[k]) / len(agg[k]) * 2048.0 / 1e6), flush=True)
PY
done
