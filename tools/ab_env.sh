#!/bin/bash
# GPU box: alternate bench runs (graph replay, no profiling) with and without an environment switch:
#   tools/ab_env.sh DMVAE_KNOBS=10=1 [rounds]
cd $GRAFT_REPO_ROOT
V=$1; R=${2:-4}
for r in $(seq $R); do
  for mode in base "$V"; do
    if [ "$mode" = base ]; then E="X_UNUSED=1"; else E="$V"; fi
    env $E python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --profile-steps 0 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s' % '$mode', d['ms_per_step'])"
  done
done
