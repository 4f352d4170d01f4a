#!/bin/bash
# Round evidence for the CNN-trunk step, run on the GPU box:  tools/profile_cnn.sh r01
#   bench.py --cnn (B = 4096, bf16)                 -> <tag>_cnn_bench.json
#   rocprofv3 --kernel-trace --stats of the same    -> <tag>_cnn_kernel_stats.csv
# Files land in gpurun_out/profiles_<tag>/ ; copy them into profiles/.
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
ARGS="--cnn --enc_layers 500 --steps 30 --warmup 5"
python3 bench.py $ARGS > $OUT/${TAG}_cnn_bench.json 2> $OUT/${TAG}_cnn_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cnn_$TAG -o $TAG -- python3 bench.py $ARGS --no-cpu-baseline --profile-steps 0 > $OUT/${TAG}_cnn_bench_under_rocprof.json 2> /dev/null
cp $(ls gpurun_out/prof_cnn_$TAG/*/*kernel_stats.csv gpurun_out/prof_cnn_$TAG/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_cnn_kernel_stats.csv
python3 tools/showbench.py $OUT/${TAG}_cnn_bench.json | head -8
head -14 $OUT/${TAG}_cnn_kernel_stats.csv
