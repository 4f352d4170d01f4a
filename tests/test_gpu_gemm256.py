"""GPU tests of the 256x256 macro-tile bf16 GEMM (csrc/gemm_bf16_256.hip) through the C ABI: every operand layout
and fused epilogue it is instantiated for, on exact small integers (bf16 operands and fp32 accumulation are exact
there, so a wrong fragment / swizzle / stagger / buffer shows as a hard mismatch) and on random data.  The kernel
is normally selected only when its grid covers the chip; `dmvae_debug_set_knob(6, 2)` selects it for every shape
that divides by 256.  The K loop is pipelined 6 half-tiles deep over two staggered wave groups: K = 64 (one K
tile), odd and even tile counts and long K are all covered, repeated to give a hand-off race a chance to show."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip256():
    import dmvae_hip
    from dmvae_hip import _lib
    torch.cuda.set_device(0)
    _lib.check(_lib.lib.dmvae_debug_set_knob(6, 2))
    yield _lib
    _lib.check(_lib.lib.dmvae_debug_set_knob(6, 1))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


def operands(layout, M, N, K, rng, integer):
    if integer:
        A = rng.randint(-3, 4, size=(M, K)).astype(np.float64)
        B = rng.randint(-3, 4, size=(K, N)).astype(np.float64)
    else:
        A, B = rng.randn(M, K), rng.randn(K, N)
    if layout == 0:
        return A, B, A, B
    if layout == 1:
        return A, B, A, B.T.copy()
    return A, B, A.T.copy(), B


def gemm(L, layout, M, N, K, A, lda, B, ldb, epi):
    L.check(L.lib.dmvae_gemm(stream(), 1, layout, M, N, K, L.ptr(A), lda, L.ptr(B), ldb, C.byref(epi), 1), "dmvae_gemm")


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(256, 256, 64), (256, 512, 128), (512, 256, 192), (768, 512, 320), (256, 256, 2048), (1024, 1280, 704)])
def test_gemm256_layouts_exact_integers(hip256, layout, shape):
    L = hip256
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K + layout)
    A, B, Am, Bm = operands(layout, M, N, K, rng, True)
    Ad, Bd = dev(Am, torch.bfloat16), dev(Bm, torch.bfloat16)
    ref = A @ B
    e = L.Epilogue()
    e.kind = L.EPI_STORE_F32
    for rep in range(3):        # the same launch repeated: a racy hand-off rarely fails twice the same way
        out = torch.full((M, N), 777.0, dtype=torch.float32, device="cuda")
        e.out, e.ldo = L.ptr(out).value, N
        gemm(L, layout, M, N, K, Ad, Am.shape[1], Bd, Bm.shape[1], e)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().astype(np.float64), ref, err_msg="repeat %d" % rep)


def test_gemm256_matches_the_smaller_tiles_on_random_data(hip256):
    """same problem through the 256x256 kernel and (knob 6 = 0) through the 128-wide tiles: both accumulate the same
    bf16 products in fp32, in a different order"""
    L = hip256
    rng = np.random.RandomState(0)
    for layout, (M, N, K) in [(0, (1024, 2048, 512)), (1, (2048, 512, 1024)), (2, (512, 1024, 4096))]:
        A, B, Am, Bm = operands(layout, M, N, K, rng, False)
        Ad, Bd = dev(Am, torch.bfloat16), dev(Bm, torch.bfloat16)
        Ar, Br = Ad.float().cpu().numpy().astype(np.float64), Bd.float().cpu().numpy().astype(np.float64)
        ref = (Ar if layout != 2 else Ar.T) @ (Br if layout != 1 else Br.T)
        outs = []
        for policy in (2, 0):
            L.check(L.lib.dmvae_debug_set_knob(6, policy))
            out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
            e = L.Epilogue()
            e.kind = L.EPI_STORE_F32
            e.out, e.ldo = L.ptr(out).value, N
            gemm(L, layout, M, N, K, Ad, Am.shape[1], Bd, Bm.shape[1], e)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        L.check(L.lib.dmvae_debug_set_knob(6, 2))
        for o in outs:
            assert np.abs(o - ref).max() < 2e-3 * math.sqrt(K), (layout, M, N, K)
        assert np.abs(outs[0] - outs[1]).max() < 1e-3 * math.sqrt(K)


def test_gemm256_fused_epilogues(hip256):
    L = hip256
    rng = np.random.RandomState(11)
    M, N, K = 512, 256, 192
    tol = 2e-2
    A, B, _, _ = operands(0, M, N, K, rng, False)
    A *= 0.2
    Ad, Bd = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    ref = Ad.float().cpu().numpy().astype(np.float64) @ Bd.float().cpu().numpy().astype(np.float64)
    bias = rng.randn(N)
    bd = dev(bias)
    # BIAS_RELU
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU; e.out, e.ldo = L.ptr(out).value, N; e.bias = L.ptr(bd).value
    gemm(L, 0, M, N, K, Ad, K, Bd, N, e)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), np.maximum(ref + bias, 0), atol=tol * 3, rtol=tol)
    # BIAS_RECON, both kinds, masked rows / columns, logits copy, loss partials on the 64x64 cell grid
    mv, nv = 500, 249
    x = rng.rand(M, N)
    xd = dev(x)
    for kind in (0, 1):
        dl = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
        lg = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        npart = L.lib.dmvae_gemm_partials(1, M, N)
        assert npart == (M // 64) * (N // 64)
        parts = torch.full((npart,), 3.0, dtype=torch.float32, device="cuda")      # every cell must be written
        e = L.Epilogue(); e.kind = L.EPI_BIAS_RECON; e.out, e.ldo = L.ptr(dl).value, N; e.bias = L.ptr(bd).value
        e.out2, e.ldo2 = L.ptr(lg).value, N
        e.aux0, e.ld0 = L.ptr(xd).value, N
        e.m_valid, e.n_valid, e.recon_kind, e.scale = mv, nv, kind, 1.0 / mv
        e.partials = L.ptr(parts).value
        gemm(L, 0, M, N, K, Ad, K, Bd, N, e)
        torch.cuda.synchronize()
        lgpu = lg.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(lgpu, ref + bias, atol=1e-3)
        mask = np.zeros((M, N)); mask[:mv, :nv] = 1
        if kind == 0:
            per = np.maximum(lgpu, 0) - lgpu * x + np.log1p(np.exp(-np.abs(lgpu)))
            dref = (1 / (1 + np.exp(-lgpu)) - x) / mv
        else:
            per = 0.5 * (lgpu - x) ** 2
            dref = (lgpu - x) / mv
        assert parts.sum().item() == pytest.approx((per * mask).sum(), rel=2e-5)
        np.testing.assert_allclose(dl.float().cpu().numpy(), dref * mask, atol=3e-5, rtol=1e-2)
    # RELU_MASK (DX layout)
    Mx, Nx, Kx = 256, 512, 192
    A2, B2, Am2, Bm2 = operands(1, Mx, Nx, Kx, rng, False)
    A2d, B2d = dev(Am2 * 0.2, torch.bfloat16), dev(Bm2, torch.bfloat16)
    ref2 = A2d.float().cpu().numpy().astype(np.float64) @ B2d.float().cpu().numpy().astype(np.float64).T
    Y = rng.randn(Mx, Nx); Y[Y < 0] = 0
    Yd = dev(Y, torch.bfloat16)
    out2 = torch.zeros((Mx, Nx), dtype=torch.bfloat16, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_RELU_MASK; e.out, e.ldo = L.ptr(out2).value, Nx; e.aux0, e.ld0 = L.ptr(Yd).value, Nx
    gemm(L, 1, Mx, Nx, Kx, A2d, Kx, B2d, Kx, e)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.float().cpu().numpy(), ref2 * (Y > 0), atol=tol * 3, rtol=tol)


def test_gemm256_weight_gradient_group_with_bias_and_fused_adam(hip256):
    """dmvae_gemm_grouped_dw / _dw_adam: the problems that divide by 256 leave the group for the macro-tile kernel
    (bias gradient from slab column sums, Adam in its epilogue), the others stay grouped; results equal the
    individual products exactly (integers) and the fused update equals gradient-then-stand-alone-Adam bit for bit."""
    _dw_group_case(hip256, 512)


def _dw_group_case(L, K):
    shapes = [(512, 768), (256, 256), (192, 256), (512, 64), (1024, 512)]
    rng = np.random.RandomState(9)
    n_el = sum(m * n + n for m, n in shapes) + 64
    arenas = {k: torch.zeros(n_el, dtype=torch.float32, device="cuda") for k in ("grad", "m", "v")}
    param = torch.as_tensor(rng.randn(n_el).astype(np.float32)).cuda()
    shadow = torch.zeros(n_el, dtype=torch.bfloat16, device="cuda")
    probs = (L.GemmProblem * len(shapes))()
    keep, refs, offs = [], [], []
    off = 0
    for i, (M, N) in enumerate(shapes):
        A, B, Am, Bm = operands(2, M, N, K, rng, True)
        Ad, Bd = dev(Am, torch.bfloat16), dev(Bm, torch.bfloat16)
        keep.append((Ad, Bd))
        refs.append((A @ B, B.sum(0)))
        offs.append((off, off + M * N))
        p = probs[i]
        p.M, p.N, p.K = M, N, K
        p.A, p.lda, p.B, p.ldb = L.ptr(Ad).value, M, L.ptr(Bd).value, N
        p.epi.kind = L.EPI_STORE_F32
        p.epi.out, p.epi.ldo = arenas["grad"].data_ptr() + 4 * off, N
        p.epi.out2 = arenas["grad"].data_ptr() + 4 * (off + M * N)
        off += M * N + N
    L.check(L.lib.dmvae_gemm_grouped_dw(stream(), 1, probs, len(shapes)), "dmvae_gemm_grouped_dw")
    torch.cuda.synchronize()
    g = arenas["grad"].cpu().numpy().astype(np.float64)
    for (M, N), (wo, bo), (ref, dbref) in zip(shapes, offs, refs):
        np.testing.assert_array_equal(g[wo:wo + M * N].reshape(M, N), ref)
        np.testing.assert_array_equal(g[bo:bo + N], dbref)
    # stand-alone Adam on the stored gradient ...
    st = L.State()
    # lr_t exactly as the device computes it (common.h adam_lr_t): float32 inputs widened to double, result rounded to float32
    lr, b1, b2 = (float(np.float32(x)) for x in (0.002, 0.9, 0.999))
    st.adam_t, st.lr, st.lr_t = 1, lr, float(np.float32(lr * math.sqrt(1.0 - b2) / (1.0 - b1)))
    state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
    p1, m1, v1, s1 = param.clone(), arenas["m"].clone(), arenas["v"].clone(), shadow.clone()
    L.check(L.lib.dmvae_adam_tf(stream(), n_el, L.ptr(p1), L.ptr(arenas["grad"]), L.ptr(m1), L.ptr(v1), L.ptr(s1), 0.0, 0.9, 0.999, 1e-8, 1.0, 0,
                                C.c_uint64(2 ** 64 - 1), L.ptr(state)), "dmvae_adam_tf")
    # ... against the update fused into the weight-gradient launches
    p2, m2, v2, s2 = param.clone(), arenas["m"].clone(), arenas["v"].clone(), shadow.clone()
    g2 = torch.zeros_like(arenas["grad"])
    for i in range(len(shapes)):
        probs[i].epi.kind = L.EPI_ADAM
        probs[i].epi.out = g2.data_ptr() + 4 * offs[i][0]
        probs[i].epi.out2 = g2.data_ptr() + 4 * offs[i][1]
    ctx = L.AdamCtx()
    ctx.param, ctx.grad, ctx.m, ctx.v, ctx.param_bf16 = p2.data_ptr(), g2.data_ptr(), m2.data_ptr(), v2.data_ptr(), s2.data_ptr()
    ctx.state, ctx.beta1, ctx.beta2, ctx.epsilon, ctx.grad_scale, ctx.store_grad = state.data_ptr(), 0.9, 0.999, 1e-8, 1.0, 0
    ctx.seg_off, ctx.seg_n = 0, 0
    L.check(L.lib.dmvae_gemm_grouped_dw_adam(stream(), probs, len(shapes), C.byref(ctx)), "dmvae_gemm_grouped_dw_adam")
    torch.cuda.synchronize()
    used = off
    for a, b, what in ((p1, p2, "param"), (m1, m2, "m"), (v1, v2, "v"), (s1, s2, "shadow")):
        assert torch.equal(a[:used], b[:used]), what
