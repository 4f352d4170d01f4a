cd /root/repo
python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_step.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/rf_t.log 2>&1; tail -3 gpurun_out/rf_t.log
tools/ab_lib.sh base --steps 300 --warmup 30 || exit 1
tools/ab_lib.sh base --config cfg3 --steps 60 --warmup 10 || exit 1
