#!/bin/bash
# How much of the data-parallel step is host issue time?  The N > 1 launch sequences with a real
# one-rank RCCL communicator (DMVAE_DP_FORCE=1) at the metric's batch and at a small batch: the GPU
# time shrinks with the batch, the host cost of issuing ~30 launches + 3 collectives does not.
#   tools/dp_host.sh            (on the GPU box; prints ms/step per mode)
B=${1:-512}
run() {   # label, env..., -- bench args
    local label=$1; shift
    env "$@" python3 bench.py --no-cpu-baseline --profile-steps 0 $ARGS > gpurun_out/.dp_host.json 2> gpurun_out/.dp_host.err \
        || { tail -5 gpurun_out/.dp_host.err; return 1; }
    python3 -c "import json; d=json.loads(open('gpurun_out/.dp_host.json').readline()); print('%-44s %s  ms/step %.4f' % ('$label', '$ARGS', d['ms_per_step']))"
}
mkdir -p gpurun_out
for ARGS in "--batch 4096" "--batch $B"; do
    run "fused single-process step (HIP graph)" X=1 &&
    run "RCCL x1, three bucket all-reduces (eager)" DMVAE_DP_FORCE=1 &&
    run "RCCL x1, one all-reduce (eager)" DMVAE_DP_FORCE=1 DMVAE_DP_OVERLAP=0 || exit 1
done
