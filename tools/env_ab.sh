#!/bin/bash
# Same-box A/B of an environment setting read at HIP start-up: alternates  env <VAR=VALUE ...> bench.py  and plain bench.py, 3 rounds.
#   tools/env_ab.sh "HIP_FORCE_DEV_KERNARG=1" [bench.py args...]
SET=${1:?"VAR=VALUE [VAR=VALUE ...]"}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
for r in 1 2 3; do
  for w in with without; do
    if [ $w = with ]; then env $SET python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --elbo-epochs 0 > "$ROOT/gpurun_out/env_${w}_$r.json" || exit 1
    else python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --elbo-epochs 0 > "$ROOT/gpurun_out/env_${w}_$r.json" || exit 1; fi
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('%-8s %.4f ms/step' % (sys.argv[2], d['ms_per_step']), flush=True)" "$ROOT/gpurun_out/env_${w}_$r.json" $w
  done
done
