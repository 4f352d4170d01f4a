#!/bin/bash
# HBM traffic of the step's kernels from PMC counters, per MI355X_MICROARCH.md "HBM" section:
# separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), kernel-trace only;
# gfx950 correction: FETCH_SIZE reports HALF of the bytes of wide coalesced reads -> doubled;
# WRITE_SIZE is exact for 16-B/lane stores.  Units: KiB.  Writes profiles/traffic.json and
# gpurun_out/pmc_traffic.txt.   Run on the GPU box:  tools/pmc_traffic.sh     (CFG=cfg3|cfg4|cfg5: that BASELINE config
# instead of the metric config -> gpurun_out/traffic_$CFG.json, gpurun_out/pmc_traffic_$CFG.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${CFG:-}
SUF=${CFG:+_$CFG}
CMD="python3 bench.py ${CFG:+--config $CFG} --steps ${PMC_STEPS:-20} --warmup 5 --repeats 1 --no-cpu-baseline --elbo-epochs 0 --profile-steps 0 --no-graph"
rm -rf gpurun_out/pmc_tr
export SUF
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_tr/f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_tr/w -- $CMD > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, json, re, os
def load(sub, ctr):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_tr/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "dmvae::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
fe, wr = load("f", "FETCH_SIZE"), load("w", "WRITE_SIZE")
out, lines = {}, []
for k in sorted(fe, key=lambda k: -sum(fe[k])):
    m = re.search(r"dmvae::(\w+(?:<[^>]*>)?)", k)
    name = m.group(1) if m else k
    f = sum(fe[k]) / len(fe[k]) * 1024.0 * 2.0          # KiB -> B, x2 (gfx950 FETCH_SIZE correction)
    w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024.0
    out[name] = {"hbm_bytes_per_launch": f + w, "fetch_bytes_corrected": f, "write_bytes": w, "launches": len(fe[k])}
    lines.append("%-60s launches %4d  fetch(x2) %10.2f MB  write %9.2f MB  total %10.2f MB/launch" % (name[:60], len(fe[k]), f / 1e6, w / 1e6, (f + w) / 1e6))
os.makedirs("gpurun_out", exist_ok=True)
suf = os.environ.get("SUF", "")
json.dump(out, open("gpurun_out/traffic%s.json" % suf, "w"), indent=1)
open("gpurun_out/pmc_traffic%s.txt" % suf, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
PY
