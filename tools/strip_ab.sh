#!/bin/bash
# VERDICT r4 #7 -- a ROW-STRIP forward measured, not estimated: the fused heads + latent launch (csrc/heads_latent.hip) is one -- a workgroup owns 16 rows, keeps their
# intermediate (the heads' f32 outputs) on the CU and streams both weight matrices from L2 -- and knob 19 = 2 lifts its one-round limit, so that at 8192 / 16 384 rows
# every CU runs 2 / 4 strips in turn.  Step A/B against the tiled pair it replaces (grouped 64-row heads GEMM + latent_fwd_kernel), then the launch's own duration.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in cfg2:batch=8192 cfg2:batch=16384 cfg3; do python3 tools/knob_step.py $c 19 0 2 2 0 2>&1 | grep -v amdgpu.ids | tail -1; done
for c in cfg2:batch=16384 cfg3; do
  for k in 0 2; do DMVAE_KNOBS="19=$k" tools/step_trace.sh $c strip_$k 2>&1 | grep -i "latent\|<64, 64, 0, 1" | sed "s/^/$c knob 19 = $k: /"; done
done
