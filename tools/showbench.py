import json,sys
d=json.load(open(sys.argv[1]))
for k in ("value","ms_per_step","step_mfma_frac_of_peak","last_loss","kernel_ms_per_step_sum"): print(k, d.get(k))
print(d.get("roofline")); print(d.get("cpu_baseline"))
for r in d.get("kernels",[]): print(r)
