"""The grouped dX of the two head layers (d(z-hidden) | d(c-hidden) = dY . W^T through the ReLU mask of [hz | hc]) alone.

    python tools/heads_dx_probe.py [cfg2|cfg3|cfg4]        (DMVAE_HIP_LIB selects an ablated build, tools/ablate.sh)

A memory-bound launch -- K = 2 D (64..512) and the padded class count (64) against 2 x B x 2048 bf16 of mask read and of output
written -- that the step runs at 2-2.8 TB/s.  Prints us per launch (graph replay of 20 launches, median of 5) and the TB/s of
mask + output for: the grouped launch under the tile-policy knobs, each problem alone, and a plain masked copy of the same bytes
(torch) as the rate a streaming kernel reaches on this box."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L

SHAPES = {"cfg2": (4096, 128, 64), "cfg3": (16384, 256, 64), "cfg4": (8192, 512, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B, D2, Kp = SHAPES[name]
H = 2048
torch.cuda.set_device(0)
side = torch.cuda.Stream()
g = torch.Generator(device="cuda"); g.manual_seed(0)
dmv = torch.randn(B, D2, device="cuda", generator=g).bfloat16()
dlg = torch.randn(B, Kp, device="cuda", generator=g).bfloat16()
Wmv = torch.randn(H, D2, device="cuda", generator=g).bfloat16()       # [in = H][out = 2 D]: contraction-contiguous for dX
Wl = torch.randn(H, Kp, device="cuda", generator=g).bfloat16()
hzc = torch.relu(torch.randn(B, 2 * H, device="cuda", generator=g)).bfloat16()
out = torch.zeros(B, 2 * H, device="cuda", dtype=torch.bfloat16)


def prob(dY, Kd, W, col):
    p = L.GemmProblem(); p.M, p.N, p.K = B, H, Kd
    p.A, p.lda, p.B, p.ldb = dY.data_ptr(), Kd, W.data_ptr(), Kd
    p.epi.kind = L.EPI_RELU_MASK; p.epi.out = out.data_ptr() + col * 2; p.epi.ldo = 2 * H
    p.epi.aux0 = hzc.data_ptr() + col * 2; p.epi.ld0 = 2 * H
    return p


pz, pc = prob(dmv, D2, Wmv, 0), prob(dlg, Kp, Wl, H)


def timeit(fn, n=20, reps=5):
    with torch.cuda.stream(side):
        st = C.c_void_p(side.cuda_stream)
        for _ in range(2): fn(st)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(n): fn(st)
        ts = []
        for _ in range(reps):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(side); gr.replay(); t1.record(side); side.synchronize()
            ts.append(t0.elapsed_time(t1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


def group(ps):
    arr = (L.GemmProblem * len(ps))(*ps)
    return lambda st: L.check(L.lib.dmvae_gemm_grouped(st, 1, L.GEMM_DX, arr, len(ps)))


def single(p):
    return lambda st: L.check(L.lib.dmvae_gemm(st, 1, L.GEMM_DX, p.M, p.N, p.K, p.A, p.lda, p.B, p.ldb, C.byref(p.epi), 1))


mb = 2.0 * B * 2 * H * 2 / 1e6          # mask read + output written
def show(what, us, frac=1.0):
    print("%-46s %7.1f us   %5.2f TB/s (mask + output)" % (what, us, mb * frac / us), flush=True)


print("%s: B=%d  K = %d | %d  N = 2 x %d   mask + output = %.0f MB   lib=%s" % (name, B, D2, Kp, H, mb, os.environ.get("DMVAE_HIP_LIB", "in-tree")))
ref = (torch.relu(dmv.float() @ Wmv.float().t()) * 0 + (dmv.float() @ Wmv.float().t()) * (hzc[:, :H] > 0)).bfloat16() if B <= 4096 else None
show("grouped [z, c] (planned tiles)", timeit(group([pz, pc])))
if ref is not None and "abl" not in os.environ.get("DMVAE_HIP_LIB", ""):
    torch.cuda.synchronize()
    err = (out[:, :H].float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
    print("   (max error vs torch: %.2e of the largest element)" % err)
for mode, label in ((0, "all 64x64"), (2, "largest tile each shape divides")):
    L.check(L.lib.dmvae_debug_set_knob(2, mode))
    show("grouped, knob 2 = %d (%s)" % (mode, label), timeit(group([pz, pc])))
L.check(L.lib.dmvae_debug_set_knob(2, 1))
L.check(L.lib.dmvae_debug_set_knob(7, 1))
show("grouped, knob 7 = 1 (64x64 / 2-slot, 4-5 per CU)", timeit(group([pz, pc])))
L.check(L.lib.dmvae_debug_set_knob(7, 0))
L.check(L.lib.dmvae_debug_set_knob(9, 8))
show("grouped, knob 9 = 8 (eight waves per workgroup)", timeit(group([pz, pc])))
L.check(L.lib.dmvae_debug_set_knob(2, 2))
show("grouped, knob 9 = 8, knob 2 = 2", timeit(group([pz, pc])))
L.check(L.lib.dmvae_debug_set_knob(2, 1))
for nw in (8, 0):
    L.check(L.lib.dmvae_debug_set_knob(9, nw))
    show("group [z] only, knob 9 = %d" % nw, timeit(group([pz])), 0.5)
    show("group [c] only, knob 9 = %d" % nw, timeit(group([pc])), 0.5)
    show("group [c, z], knob 9 = %d" % nw, timeit(group([pc, pz])))
    L.check(L.lib.dmvae_debug_set_knob(4, 0))
    show("group [z, c], knob 4 = 0 (XCD runs per problem), knob 9 = %d" % nw, timeit(group([pz, pc])))
    L.check(L.lib.dmvae_debug_set_knob(4, 1))
    for gm in (1, 4, 16, 128):
        L.check(L.lib.dmvae_debug_set_knob(0, gm))
        show("group [z, c], supertile rows %d, knob 9 = %d" % (gm, nw), timeit(group([pz, pc])))
    L.check(L.lib.dmvae_debug_set_knob(0, 0))
show("z alone (stand-alone kernel)", timeit(single(pz)), 0.5)
show("c alone (stand-alone kernel)", timeit(single(pc)), 0.5)
L.check(L.lib.dmvae_debug_set_knob(6, 2))
show("z alone, knob 6 = 2 (256x256 macro tile)", timeit(single(pz)), 0.5)
L.check(L.lib.dmvae_debug_set_knob(6, 1))
show("torch: out = where(hzc > 0, out2, 0) (same bytes + 1 read)", timeit(lambda st: torch.where(hzc > 0, hzc, hzc, out=out)), 1.5)
show("torch: out.copy_(hzc)  (read + write)", timeit(lambda st: out.copy_(hzc)))
