"""Builds libdmvae_hip.so (gfx950) in-tree with hipcc.  `python build.py [-f]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "dmvae_hip", "libdmvae_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_bf16_256.hip", "gemm_f32.hip", "latent.hip", "elementwise.hip", "conv.hip", "api.hip"]
# -amdgpu-mfma-vgpr-form: accumulators stay in VGPRs.  Left to its default, hipcc (ROCm 7.2) puts the
# 4-wave GEMM tiles' accumulators in AGPRs and then shuffles them through v_accvgpr_read/write/mov on
# every K step (256 such moves against 80 MFMAs in the 128x128 dW loop); no kernel here needs > 256 VGPRs.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "dmvae_hip.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            print(r.stderr[-3000:], file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
