#!/bin/bash
# round 4, GPU call B: the streaming heads-dX kernel (second form): tests, the classification diagnostic, then -- only if the tests pass -- timings
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out

timeout -k 10 120 python tools/heads_dx_diag.py 1 4096 2>&1 | tee $O/r04_hdx_diag_knob1.txt | cut -c1-420
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "heads_dx" > $O/r04_hdx_tests.log 2>&1
rc=$?; tail -3 $O/r04_hdx_tests.log
if [ $rc -ne 0 ]; then echo "heads_dx tests FAILED (rc=$rc): no timing runs"; grep -E "AssertionError: \(" $O/r04_hdx_tests.log | head -20; exit 1; fi
timeout -k 10 200 python tools/heads_dx_ab.py cfg2 cfg3 cfg4 2>&1 | tee $O/r04_hdx_ab.txt &&
for cfg in cfg2 cfg3 cfg4; do timeout -k 10 200 python tools/knob_step.py $cfg 13 0 1 2>&1 | tee -a $O/r04_hdx_step_ab.txt || exit 1; done
deep-mixture-vae_amd/build/lds_probe | tee $O/r04_lds_probe2.txt
timeout -k 10 200 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "full_size_bf16_gradients" 2>&1 | grep -E "cfg[234]:|passed|failed" | tee $O/r04_fullsize_figures.txt
