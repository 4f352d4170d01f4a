"""Diagnostic for csrc/heads_dx.hip: where the streaming kernel's output differs from the grouped tiles', and what the wrong cells hold.
    python tools/heads_dx_diag.py <knob 13 value: 1 | 2> [M]
For every 64 x 128 cell that differs it prints which step of its workgroup's walk the cell belongs to and how well the kernel's cell
agrees with four candidates: the right answer, the PREVIOUS row tile's dY under this tile's gates (a stale dY block: the slot was read
before the DMA landed), this tile's dY under the previous tile's gates (gates consumed before they landed), the NEXT tile's dY (the slot
overwritten before every wave had read it)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
knob = int(sys.argv[1]); M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N, Ks = 2048, (128, 64)
torch.cuda.set_device(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tiles = M // 64
chunks = max(1, min(tiles, (512 + 16) // 32))          # the launch arithmetic of heads_dx_stream_launch for these two problems (32 slices)
steps = -(-tiles // chunks)
if steps < 2 and tiles >= 2: steps = 2
bad_launches = 0
SEEDS = 12
for seed in range(SEEDS):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    dY = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for K in Ks]
    W = [(torch.randn(N, K, device="cuda", generator=g) * 0.25).bfloat16() for K in Ks]
    gate = torch.relu(torch.randn(M, 2 * N, device="cuda", generator=g)).bfloat16()
    outs = {}
    for kb in (0, knob):
        L.check(L.lib.dmvae_debug_set_knob(13, kb))
        out = torch.zeros(M, 2 * N, device="cuda", dtype=torch.bfloat16)
        probs = (L.GemmProblem * 2)()
        for i in range(2):
            p = probs[i]; p.M, p.N, p.K = M, N, Ks[i]
            p.A, p.lda, p.B, p.ldb = dY[i].data_ptr(), Ks[i], W[i].data_ptr(), Ks[i]
            p.epi.kind = L.EPI_RELU_MASK; p.epi.out = out.data_ptr() + i * N * 2; p.epi.ldo = 2 * N
            p.epi.aux0 = gate.data_ptr() + i * N * 2; p.epi.ld0 = 2 * N
        L.check(L.lib.dmvae_gemm_grouped(st(), 1, L.GEMM_DX, probs, 2)); torch.cuda.synchronize()
        outs[kb] = out
    L.check(L.lib.dmvae_debug_set_knob(13, 1))
    a, b = outs[0], outs[knob]
    bad = (a != b)
    if not bool(bad.any()):
        continue
    bad_launches += 1
    cells = bad.view(M // 64, 64, 2 * N // 128, 128).any(dim=3).any(dim=1)          # [row tile][128-column cell]
    idx = cells.nonzero().tolist()
    by_step = {}
    for rt, cc in idx:
        by_step[rt % steps] = by_step.get(rt % steps, 0) + 1
    print("seed %d: %d wrong elements in %d cells of 64 x 128; cells by step index within the chunk (of %d): %s; by problem: z %d, c %d"
          % (seed, int(bad.sum()), len(idx), steps, dict(sorted(by_step.items())), sum(1 for _, c in idx if c < 16), sum(1 for _, c in idx if c >= 16)))
    for rt, cc in idx[:5]:
        i = 0 if cc < 16 else 1
        col = slice((cc % 16) * 128, (cc % 16) * 128 + 128)
        gcol = slice(cc * 128, cc * 128 + 128)

        def ref(rows, gate_rows):
            return ((dY[i][rows].float() @ W[i][col].float().t()) * (gate[gate_rows, gcol].float() > 0)).bfloat16()
        rows = slice(rt * 64, rt * 64 + 64)
        cand = {"right answer": ref(rows, rows)}
        if rt > 0:
            prev = slice(rt * 64 - 64, rt * 64)
            cand["stale dY block (previous tile's dY)"] = ref(prev, rows)
            cand["stale gates (previous tile's gates)"] = ref(rows, prev)
        if rt + 1 < tiles:
            cand["next tile's dY (slot overwritten early)"] = ref(slice(rt * 64 + 64, rt * 64 + 128), rows)
        got = b[rows, gcol]
        nz = (cand["right answer"] != 0) | (got != 0)                    # compare where something is non-zero
        match = {k: round(float(((got == v) & nz).sum() / max(1, int(nz.sum()))), 3) for k, v in cand.items()}
        print("   row tile %3d = step %d, column cell %2d (problem %s): %4d of 8192 elements wrong; fraction of its non-zero elements equal to -> %s"
              % (rt, rt % steps, cc, "zc"[i], int(bad[rows, gcol].sum()), match))
print("knob 13 = %d, M = %d (%d steps per workgroup): %d of %d launches differ from the grouped tiles" % (knob, M, steps, bad_launches, SEEDS))
