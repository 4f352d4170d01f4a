"""Run under the AddressSanitizer + UBSan host build (tests/test_host.py::test_plan_layout_under_address_sanitizer):
plan creation / sizes / tensor table / destruction for every BASELINE.json geometry and the CNN trunk, the
argument-error paths, and the tile / latent geometry helpers -- host code only, no GPU call."""
import ctypes as C
import sys

from dmvae_hip import _lib

CONFIGS = [
    dict(I=784, D=10, K=10, enc=(500, 500), head=2000, dec=(2000, 500, 500), B=100, dt=_lib.F32, trunk=0),
    dict(I=784, D=64, K=10, enc=(500, 500), head=2000, dec=(2000, 500, 500), B=4096, dt=_lib.BF16, trunk=0),
    dict(I=784, D=128, K=10, enc=(500, 500), head=2000, dec=(2000, 500, 500), B=16384, dt=_lib.BF16, trunk=0),
    dict(I=784, D=256, K=50, enc=(500, 500), head=2000, dec=(2000, 500, 500), B=8192, dt=_lib.BF16, trunk=0),
    dict(I=4096, D=512, K=256, enc=(4096,) * 4, head=4096, dec=(4096,) * 4, B=8192, dt=_lib.BF16, trunk=0),
    dict(I=784, D=64, K=10, enc=(500,), head=2000, dec=(2000, 500, 500), B=256, dt=_lib.BF16, trunk=1),
    dict(I=96, D=8, K=4, enc=(64,) * 8, head=64, dec=(64,) * 8, B=37, dt=_lib.F32, trunk=0),
    dict(I=784, D=10, K=10, enc=(128,), head=64, dec=(500, 500, 2000), B=128, dt=_lib.BF16, trunk=1, model=1),      # VaDE(cnn=True)
]
n_plans = 0
for c in CONFIGS:
    cfg = _lib.Config()
    cfg.input_dim, cfg.latent_dim, cfg.n_classes = c["I"], c["D"], c["K"]
    cfg.n_enc, cfg.head_dim, cfg.n_dec = len(c["enc"]), c["head"], len(c["dec"])
    for i, v in enumerate(c["enc"]):
        cfg.enc[i] = v
    for i, v in enumerate(c["dec"]):
        cfg.dec[i] = v
    cfg.dtype, cfg.max_batch, cfg.trunk, cfg.model = c["dt"], c["B"], c["trunk"], c.get("model", 0)
    cfg.beta1, cfg.beta2, cfg.adam_eps = 0.9, 0.999, 1e-8
    h = C.c_void_p()
    _lib.check(_lib.lib.dmvae_plan_create(C.byref(cfg), C.byref(h)), "dmvae_plan_create")
    sz = _lib.Sizes()
    _lib.check(_lib.lib.dmvae_plan_sizes(h, C.byref(sz)), "dmvae_plan_sizes")
    assert sz.param_elems > 0 and sz.work_bytes > 0 and sz.batch_pad >= c["B"] and sz.n_tensors >= 2 * (len(c["enc"]) + len(c["dec"]) + (3 if c.get("model") else 6)) + 2
    end = 0
    for i in range(sz.n_tensors):
        ti = _lib.TensorInfo()
        _lib.check(_lib.lib.dmvae_plan_tensor(h, i, C.byref(ti)), "dmvae_plan_tensor")
        assert 0 <= ti.offset and ti.offset + (ti.rows - 1) * ti.ld + ti.cols <= sz.param_elems, ti.name
        end = max(end, ti.offset + (ti.rows - 1) * ti.ld + ti.cols)
    assert end <= sz.param_elems
    ti = _lib.TensorInfo()
    assert _lib.lib.dmvae_plan_tensor(h, sz.n_tensors, C.byref(ti)) == -1          # index past the table
    b = (C.c_int64 * 5)()
    _lib.check(_lib.lib.dmvae_plan_grad_buckets(h, b), "dmvae_plan_grad_buckets")
    assert 0 == b[0] <= b[1] <= b[2] <= b[3] < b[4] == sz.param_elems and b[3] % 4096 == 0      # weight buckets, then the tail (biases, prior tables)
    assert _lib.lib.dmvae_plan_update(h, None, 1.0) != 0                            # not bound: an error, not a crash
    _lib.lib.dmvae_plan_destroy(h)
    n_plans += 1
    # geometry helpers on this config
    assert _lib.lib.dmvae_latent_nblocks(sz.batch_pad, c["D"], c["K"]) >= 1
    assert _lib.lib.dmvae_latent_ws_bytes(sz.batch_pad, c["D"], c["K"], 0) >= 0
    assert _lib.lib.dmvae_gemm_partials(1, sz.batch_pad, sz.input_pad) == (sz.batch_pad // 64) * (sz.input_pad // 64)
# argument errors
bad = _lib.Config()
h = C.c_void_p()
assert _lib.lib.dmvae_plan_create(C.byref(bad), C.byref(h)) == -1 and b"bad dims" in _lib.lib.dmvae_last_error()
bad.input_dim, bad.latent_dim, bad.n_classes, bad.head_dim, bad.n_enc, bad.n_dec, bad.max_batch = 10, 2, 2, 8, 9, 1, 4
assert _lib.lib.dmvae_plan_create(C.byref(bad), C.byref(h)) == -1
assert _lib.lib.dmvae_plan_create(None, C.byref(h)) == -1
print("asan probe ok: %d plans" % n_plans)
sys.exit(0)
