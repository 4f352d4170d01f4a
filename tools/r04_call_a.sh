#!/bin/bash
# round 4, GPU call A: suite on the current build, printed figures of the full-size parity test, LDS read rates,
# data-parallel launch sequences priced with a one-rank RCCL communicator (1 / 2 / 3 buckets), relaxed-mode bench line
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/r04_t3.log 2>&1; echo "suite rc=$?"; tail -2 $O/r04_t3.log
timeout -k 10 200 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "full_size_bf16_gradients" 2>&1 | grep -E "^cfg|passed|failed" | tee $O/r04_fullsize_figures.txt
deep-mixture-vae_amd/build/lds_probe | tee $O/r04_lds_probe.txt
for cfg in cfg2 cfg4; do
  for form in "0 3" "1 2" "1 3"; do
    set -- $form
    DMVAE_DP_FORCE=1 DMVAE_DP_OVERLAP=$1 DMVAE_DP_BUCKETS=$2 timeout -k 10 120 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --elbo-epochs 0 --profile-steps 0 \
      > $O/r04_dp_${cfg}_ov$1_b$2.json 2> $O/r04_dp_${cfg}_ov$1_b$2.err || echo "dp $cfg $form FAILED"
  done
  timeout -k 10 120 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --elbo-epochs 0 --profile-steps 0 > $O/r04_dp_${cfg}_single.json 2> $O/r04_dp_${cfg}_single.err
done
timeout -k 10 120 python bench.py --mode relaxed --steps 200 --warmup 20 --no-cpu-baseline --elbo-epochs 0 > $O/r04_relaxed.json 2> $O/r04_relaxed.err || echo "relaxed FAILED"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_dp_*.json")) + ["gpurun_out/r04_relaxed.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], d.get("exchange"), d["config"].get("update", "")[:60])
    except Exception as e:
        print(f, "unreadable:", e)
PY
