"""Builds libdmvae_hip.so (gfx950) in-tree with hipcc.  `python build.py [-f]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "dmvae_hip", "libdmvae_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_bf16_256.hip", "gemm_f32.hip", "latent.hip", "latent_mfma.hip", "latent_vade.hip", "elementwise.hip", "conv.hip", "heads_dx.hip", "heads_latent.hip", "strip_fwd2.hip", "api.hip"]
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
    headers += [os.path.join(HERE, "..", "include", h) for h in ("dmvae_hip.h", "dmvae_hip_debug.h")]
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


def build_host_asan(verbose=False):
    """Host-only build of the same sources with AddressSanitizer + UBSan (hipcc --cuda-host-only: no device code, so
    no GPU sanitizer is involved and nothing here runs on a GPU box): build/asan/libdmvae_hip_asan.so.  It exercises the
    host-only parts of the library -- plan creation / arena layout / tensor table / argument checks / tile planner --
    under tests/test_host.py::test_plan_layout_under_address_sanitizer.  Returns (library, asan runtime to LD_PRELOAD)."""
    import glob
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build", "asan")
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(objdir, "libdmvae_hip_asan.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", h) for h in ("dmvae_hip.h", "dmvae_hip_debug.h")]
    flags = ["--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-fPIC", "-std=c++17", "-fsanitize=address,undefined",
             "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    objs, jobs = [], []
    for src in SOURCES:
        sp, op = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        if _stale(op, [sp] + headers):
            jobs.append([hipcc] + flags + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (host asan) failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or _stale(out, objs):
        # a host-only object still registers its (absent) device code object, __hip_fatbin_<hash>: give every one an EMPTY
        # clang offload bundle (magic + zero entries), which the HIP runtime registers and never looks into
        und = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
        syms = sorted({w for line in und.splitlines() for w in line.split() if w.startswith("__hip_fatbin_")})
        stub = os.path.join(objdir, "fatbin_stubs.c")
        with open(stub, "w") as f:
            for sname in syms:
                f.write('const char %s[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";\n' % sname)
        run(["gcc", "-c", "-fPIC", stub, "-o", stub.replace(".c", ".o")])
        run([hipcc, "--offload-arch=gfx950", "--cuda-host-only", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", out]
            + objs + [stub.replace(".c", ".o")])
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return out, (rt[-1] if rt else None)


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
