#!/bin/bash
# Same-box A/B of ONE launch -- the cfg2 weight-gradient group + Adam (gemm_bf16_grouped_kernel<64, 64, 2, 8, ...>) -- between two TREES, as rocprofv3 sees it
# in replay (67.5 us in round 3's evidence, 70.5 us in round 4's, on different boxes).
#   tools/dw_launch_ab.sh <other tree, relative to the repo root> [rounds]
# Alternates  rocprofv3 --kernel-trace --stats -- python3 <tree>/bench.py --steps 200 --warmup 20  over both trees; prints the launch's average / min
# per run, then the same on THIS tree with the things round 4 put into or beside that grid switched off one at a time (DMVAE_KNOBS, read by bench.py).
set -e
OTHER=${1:?path of the other tree}; ROUNDS=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/dw_ab; mkdir -p $O
one() {   # tag, tree, [env assignment]
    local tag=$1 tree=$2 knobs=$3
    ( cd $tree && DMVAE_KNOBS="$knobs" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o $tag -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --elbo-epochs 0 > $O/$tag.json 2> $O/$tag.err ) || { tail -5 $O/$tag.err; return 1; }
    local csv=$(ls $O/prof_$tag/*/*kernel_stats.csv $O/prof_$tag/*kernel_stats.csv 2>/dev/null | head -1)
    python3 - "$csv" "$O/$tag.json" "$tag" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
dw = [r for r in rows if "grouped_kernel<64, 64, 2, 8" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"])
ms = json.load(open(sys.argv[2]))["ms_per_step"]
for r in dw:
    print("%-14s dW+Adam launch: avg %.2f us  min %.2f  max %.2f  (%s calls)   step %.4f ms" % (sys.argv[3], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
          float(r["MaxNs"]) / 1e3, r["Calls"], ms), flush=True)
PY
    rm -rf $O/prof_$tag
}
for r in $(seq 1 $ROUNDS); do
    one other_$r $ROOT/$OTHER ""
    one this_$r $ROOT ""
done
# what round 4 put into / beside the grid, one at a time (knob list: include/dmvae_hip_debug.h)
one fin_own $ROOT "16=0"          # step_finalize as a launch of its own (no riders anywhere)
one fin_heads $ROOT "16=2"        # riders in the heads' dX launch (round 3's place)
one thin_off $ROOT "18=0"         # dZ GEMM on the general tiles
one heads_grp $ROOT "13=0"        # heads' dX on the grouped tiles (round 3's kernel)
one this_last $ROOT ""
