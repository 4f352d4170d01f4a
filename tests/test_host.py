"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol the
header declares, host-side logic of the drop-in (samplers, Dataset, CLI flags,
sharding) against the reference's golden vectors, and the data-parallel
exchange on a world_size-2 gloo group.  No GPU compute is called here."""
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from dmvae_hip import _lib
    declared = set()
    for h in ("dmvae_hip.h", "dmvae_hip_debug.h"):      # the drop-in boundary, and the measurement / tuning entries
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names = set(re.findall(r"\b(dmvae_[a-z0-9_]+)\s*\(", hdr))
        assert all(("debug" in n or "prof" in n) == (h == "dmvae_hip_debug.h") for n in names), (h, names)
        declared |= names
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), "libdmvae_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    m = re.search(r"#define DMVAE_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "dmvae_hip.h")).read())
    assert _lib.lib.dmvae_abi_version() == int(m.group(1)) == _lib.ABI_VERSION == 5


def test_ctypes_structs_match_header_layout():
    import ctypes as C
    from dmvae_hip import _lib
    # sizes computed from the C declarations (natural alignment, LP64)
    assert C.sizeof(_lib.State) == 8 + 8 + 4 + 4 + 4 * 4 + 4 * 4 + 4 * 4
    assert C.sizeof(_lib.Epilogue) == 24 + 12 * 8
    assert C.sizeof(_lib.Buffers) == 8 * 8
    assert C.sizeof(_lib.TensorInfo) == 32 + 8 + 4 + 4 + 8
    assert C.sizeof(_lib.ProfRow) == 48 + 8 + 8 + 8 + 8 + 8 + 8
    assert C.sizeof(_lib.Config) == 4 * 3 + 4 + 32 + 4 + 4 + 32 + 4 * 4 + 4 * 4 + 8 + 4 + 4 + 4 + 4


def test_ctypes_structs_match_the_header_as_a_c_compiler_lays_it_out(tmp_path):
    """The boundary is a C header: compile it with gcc (plain C, no HIP) and compare sizeof and the offset of EVERY field of every public struct with the
    ctypes mirror the Python side binds (dmvae_hip/_lib.py) -- a field added on one side only, or in another order, fails here, not on the GPU."""
    import ctypes as C
    import subprocess
    from dmvae_hip import _lib
    pairs = [("dmvae_epilogue", _lib.Epilogue), ("dmvae_gemm_problem", _lib.GemmProblem), ("dmvae_adam_ctx", _lib.AdamCtx), ("dmvae_latent_args", _lib.LatentArgs),
             ("dmvae_heads_args", _lib.HeadsArgs), ("dmvae_state", _lib.State), ("dmvae_config", _lib.Config), ("dmvae_tensor_info", _lib.TensorInfo),
             ("dmvae_sizes", _lib.Sizes), ("dmvae_buffers", _lib.Buffers), ("dmvae_prof_row", _lib.ProfRow)]
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dmvae_hip.h")).read() + open(os.path.join(ROOT, "include", "dmvae_hip_debug.h")).read(), flags=re.S)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dmvae_hip_debug.h"', 'int main(void) {']
    want = {}
    for cname, cls in pairs:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):                     # "float* gmu; float* glv" and "int32_t B, B_pad" forms
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip())
                fields.append(m.group(1))
        py = [f[0] for f in cls._fields_]
        assert fields == py, (cname, [a for a in fields if a not in py], [b for b in py if b not in fields], fields, py)
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f in fields:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f))
        lines.append('printf("\\n");')
        want[cname] = [C.sizeof(cls)] + [getattr(cls, f).offset for f in py]
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in out.splitlines()}
    assert got == want, {k: (got.get(k), want[k]) for k in want if got.get(k) != want[k]}


def test_plan_layout_without_gpu():
    """plan creation is host-only: arena layout, padding and the tensor table."""
    import ctypes as C
    from dmvae_hip import _lib
    cfg = _lib.Config()
    cfg.input_dim, cfg.latent_dim, cfg.n_classes = 784, 64, 10
    cfg.n_enc, cfg.head_dim, cfg.n_dec = 2, 2000, 3
    cfg.enc[0] = cfg.enc[1] = 500
    cfg.dec[0], cfg.dec[1], cfg.dec[2] = 2000, 500, 500
    cfg.dtype, cfg.max_batch = _lib.BF16, 4096
    cfg.beta1, cfg.beta2, cfg.adam_eps = 0.9, 0.999, 1e-8
    h = C.c_void_p()
    _lib.check(_lib.lib.dmvae_plan_create(C.byref(cfg), C.byref(h)))
    sz = _lib.Sizes()
    _lib.check(_lib.lib.dmvae_plan_sizes(h, C.byref(sz)))
    assert sz.batch_pad == 4096 and sz.input_pad == 832 and sz.n_tensors == 24
    names = {}
    for i in range(sz.n_tensors):
        ti = _lib.TensorInfo()
        _lib.check(_lib.lib.dmvae_plan_tensor(h, i, C.byref(ti)))
        names[ti.name.decode()] = (ti.offset, ti.rows, ti.cols, ti.ld)
    assert names["W_enc0"][1:] == (784, 500, 512) and names["W_zh"][3] == names["W_ch"][3] == 4096
    assert names["W_ch"][0] - names["W_zh"][0] == 2048 and names["W_logvar"][0] - names["W_mean"][0] == 64
    assert names["prior_log_vars"][0] - names["prior_means"][0] == 640
    logical = sum(r * c for (_, r, c, _) in names.values())
    assert logical == 4697868 - 0 or abs(logical - 4.698e6) < 2e3      # P = 4.698 M (cfg2)
    assert sz.param_elems % 4 == 0 and sz.param_elems < 1.08 * logical   # padding overhead < 8 %
    bad = _lib.Config()
    assert _lib.lib.dmvae_plan_create(C.byref(bad), C.byref(h)) == -1
    _lib.lib.dmvae_plan_destroy(h)


def test_plan_layout_under_address_sanitizer():
    """the host-only code of the library (plan creation, arena layout, tensor table, argument checks,
    geometry helpers) built with -fsanitize=address,undefined (hipcc --cuda-host-only: no device code, CPU only) and
    driven through the C ABI in a child process with the ASan runtime preloaded.  Any heap / stack / UB report fails it."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("dmvae_build", os.path.join(ROOT, "deep-mixture-vae_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    so, rt = mod.build_host_asan()
    assert os.path.exists(so) and rt, (so, rt)
    env = dict(os.environ, LD_PRELOAD=rt, DMVAE_HIP_LIB=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=os.path.join(ROOT, "deep-mixture-vae_amd"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "asan_plan_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan probe ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dmvae_hip import StepEngine
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        StepEngine(784, 10, 10)
    src = ""
    for d, ds, fs in os.walk(os.path.join(ROOT, "deep-mixture-vae_amd")):
        ds[:] = [x for x in ds if x != "build"]          # (object files, variant libraries, other trees built for A/B runs: not the product)
        for f in fs:
            if f.endswith(".py"):
                src += open(os.path.join(d, f)).read()
    assert "dmvae_oracle" not in src and "import oracle" not in src      # the product never touches oracle/


def test_host_samplers_follow_reference_stream(golden):
    import priors
    for ci in range(int(golden["n_cases"])):
        pre = "c%d_s0_" % ci
        B, D, K = golden[pre + "shape"]
        np.random.seed(int(golden[pre + "np_seed"]))
        g = priors.DiscreteFactorial("cluster", 1, int(K)).sample_reparametrization_variable(int(B))
        mix = priors.NormalMixtureFactorial("representation", int(D), int(K))     # draws its table init first in TF too;
        np.random.seed(int(golden[pre + "np_seed"]))                               # re-seed: compare the samplers only
        g = priors.DiscreteFactorial("cluster", 1, int(K)).sample_reparametrization_variable(int(B))
        eps = mix.sample_reparametrization_variable(int(B))
        np.testing.assert_array_equal(g, golden[pre + "gumbel"])
        np.testing.assert_array_equal(eps, golden[pre + "eps"])
    from includes.utils import sample_gumbel
    np.random.seed(5)
    np.testing.assert_array_equal(sample_gumbel((7, 1, 4)), golden["gumbel_seed5"])


def test_dataset_matches_reference_epoch_semantics(golden):
    from includes.utils import Dataset, get_clustering_accuracy
    N, Bsz = 23, 5
    data = np.arange(N, dtype=np.float64)[:, None] * np.ones((1, 3))
    classes = np.arange(N) % 4
    np.random.seed(11)
    ds = Dataset((data, classes), batch_size=Bsz)
    assert ds.epoch_len == int(golden["ds_epoch_len"])
    for ep in range(2):
        batches = list(ds.get_batches())
        order = np.concatenate([b[:, 0] for b in batches]).astype(np.int64)
        np.testing.assert_array_equal(order, golden["ds_order_ep%d" % ep])
        np.testing.assert_array_equal(ds.data[:, 0].astype(np.int64), golden["ds_order_ep%d" % ep])
        np.testing.assert_array_equal(ds.classes, classes[order])
        assert [len(b) for b in batches] == list(golden["ds_batch_sizes"])
    assert get_clustering_accuracy(golden["acc_weights"], golden["acc_classes"]) == pytest.approx(float(golden["acc_value"]), abs=1e-15)


def test_cli_keeps_every_reference_flag():
    sys.argv = ["train.py"]
    import importlib
    train = importlib.import_module("train")
    ref_flags = ["model", "model_name", "dataset", "latent_dim", "output_dim", "n_clusters", "n_experts", "classification",
                 "n_epochs", "pretrain_epochs_vae", "pretrain_epochs_prior", "init_lr", "decay_rate", "decay_epochs",
                 "pretrain", "pretrain_vae_lr", "pretrain_decay_rate", "pretrain_decay_epochs", "pretrain_prior_lr",
                 "kl_annealing", "anneal_step", "anneal_epochs", "plotting", "plot_epochs", "save_epochs", "debug",
                 "visdom", "featLearn"]
    a = train.parser.parse_args([])
    for f in ref_flags:
        assert hasattr(a, f), f
    # reference defaults (code/train.py:28-97)
    assert (a.model, a.dataset, a.latent_dim, a.n_clusters, a.n_epochs, a.init_lr) == ("dmvae", "mnist", 10, -1, 500, 0.002)
    assert (a.decay_rate, a.decay_epochs, a.anneal_step, a.anneal_epochs, a.save_epochs) == (0.9, 25, 0.1, 1000, 10)
    assert a.batch_size == 100 and not a.gumbel and not a.host_noise


def test_layer_descriptors_carry_the_reference_shapes():
    """includes/layers.py + includes/network.py as descriptors: the spec lists of base_models.py:181-202 (CNN encoder) and
    :280-288 (decoder) build, expose the variable shapes the reference creates (layers.py:24-28, :45-51) and reduce to the
    conv stack / dense widths the step plan implements; anything the plan has no kernel for is refused."""
    from includes.network import DeepNetwork
    from includes import layers as Ly
    spec = [("cn", {"n_kernels": 32, "prev_n_kernels": 1, "kernel": (3, 3)}), ("cn", {"n_kernels": 32, "prev_n_kernels": 32, "kernel": (3, 3)}),
            ("mp", {"k": 2}), ("cn", {"n_kernels": 64, "prev_n_kernels": 32, "kernel": (3, 3)}),
            ("cn", {"n_kernels": 64, "prev_n_kernels": 64, "kernel": (3, 3)}), ("mp", {"k": 2}),
            ("cn", {"n_kernels": 128, "prev_n_kernels": 64, "kernel": (3, 3)}), ("cn", {"n_kernels": 128, "prev_n_kernels": 128, "kernel": (3, 3)}),
            ("mp", {"k": 2}), ("fc", {"input_dim": 2048, "output_dim": 500})]
    net = DeepNetwork("layers", spec, activation="relu", initializer="xavier")
    assert [type(l).__name__ for l in net.layers] == ["Convolution", "Convolution", "MaxPooling"] * 3 + ["FullyConnected"]
    assert net.layers[0].weight_shape == (3, 3, 1, 32) and net.layers[0].bias_shape == (32,) and net.layers[0].strides == [1, 1, 1, 1]
    assert net.layers[2].ksize == [1, 2, 2, 1] and net.layers[2].strides == [1, 2, 2, 1]
    assert net.layers[-1].weight_shape == (2048, 500) and net.layers[-1].bias_shape == (1, 500)
    stack, side = net.conv_stack(28)
    assert stack == ((1, 32, 28, False), (32, 32, 28, True), (32, 64, 14, False), (64, 64, 14, True), (64, 128, 7, False), (128, 128, 7, True))
    assert side == 4 and side * side * 128 == 2048 and net.widths() == (500,)
    dec = DeepNetwork("layers", [("fc", {"input_dim": 10, "output_dim": 2000}), ("fc", {"input_dim": 2000, "output_dim": 500}),
                                 ("fc", {"input_dim": 500, "output_dim": 500})])
    assert dec.widths() == (2000, 500, 500)
    with pytest.raises(ValueError):
        DeepNetwork("l", [("fc", {"input_dim": 10, "output_dim": 20}), ("fc", {"input_dim": 21, "output_dim": 5})]).widths()
    with pytest.raises(NotImplementedError):
        DeepNetwork("l", [("cn", {"n_kernels": 8, "prev_n_kernels": 1, "kernel": (5, 5)})]).conv_stack()
    with pytest.raises(NotImplementedError):
        DeepNetwork("l", [("bn", {"is_training": True})])
    with pytest.raises(NotImplementedError):
        DeepNetwork("l", [("xx", {})])


def test_load_data_synthetic_standin_shapes():
    from includes.utils import load_data
    ds = load_data("mnist", n_train=300, n_test=100)
    assert ds.input_dim == 784 and ds.input_type == "binary" and ds.n_classes == 10
    assert ds.train_data.shape == (300, 784) and ds.test_data.shape == (100, 784)
    assert 0.0 <= ds.train_data.min() and ds.train_data.max() < 1.0
    with pytest.raises(NotImplementedError):
        load_data("reuters")


def test_shard_range_partitions():
    from dmvae_hip import shard_range
    for n in (0, 1, 7, 100, 4096, 65000):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    for p in (os.path.join(ROOT, "deep-mixture-vae_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dmvae_oracle as O
    from dmvae_hip import GradExchange, shard_range
    cfg = O.Config(32, 4, 3, (12, 10), 14, (14, 10, 9))
    p = O.init_params(cfg, 5)
    rng = np.random.RandomState(5)
    X, eps = rng.rand(8, 32), rng.randn(8, 4)
    lo, hi = shard_range(8, rank, world)
    g = O.backward(p, cfg, O.forward(p, cfg, X[lo:hi], eps[lo:hi]))        # this rank's shard, local mean
    names = O.param_names(cfg)
    flat = torch.cat([torch.as_tensor(g[k]).reshape(-1) for k in names])     # the flat gradient arena
    ex = GradExchange()
    assert ex.enabled and ex.world == world and ex.grad_scale == 1.0 / world
    bucketed = flat.clone()
    ex(flat)                                                                 # ONE all-reduce(SUM) per step
    flat = flat * ex.grad_scale                                              # folded into the Adam kernel on the GPU
    # the overlapped form: three contiguous buckets, last part of the arena first (decoder, heads, trunk)
    assert ex.overlap
    n = bucketed.numel()
    handles = [ex.start(bucketed[lo_:hi_]) for lo_, hi_ in ((2 * n // 3, n), (n // 3, 2 * n // 3), (0, n // 3))]
    ex.finish(handles)
    assert torch.equal(bucketed * ex.grad_scale, flat)
    full = O.backward(p, cfg, O.forward(p, cfg, X, eps))
    ref = torch.cat([torch.as_tensor(full[k]).reshape(-1) for k in names])
    err = (flat - ref).abs().max().item()
    par = torch.arange(10, dtype=torch.float32) * (rank + 1)
    ex.broadcast_(par, src=0)
    m = ex.mean_scalars([float(rank), 2.0])
    out.put((rank, err, par.tolist() == list(range(10)), m))
    dist.destroy_process_group()


def test_data_parallel_exchange_world2_gloo():
    """N ranks x B/N == 1 rank x B: the all-reduce + 1/world scale
    reproduces the full-batch gradient of the oracle on a 2-rank gloo group."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, bc_ok, m in res:
        assert err < 1e-12, (rank, err)
        assert bc_ok
        assert m == pytest.approx([0.5, 2.0])


def test_epoch_plan_gives_every_rank_the_same_step_count():
    """the ragged tail of an epoch must not give one rank an extra step (its gradient all-reduce would
    pair with another rank's loss all-reduce).  With world > 1 only full global batches are trained."""
    from dmvae_hip.parallel import epoch_plan
    order = np.random.RandomState(0).permutation(1999)
    plans = [epoch_plan(order, 1000, r, 2) for r in range(2)]           # N % B = 999 > B - world
    assert [p[1] for p in plans] == [1, 1] and [p[2] for p in plans] == [0, 0]
    assert [len(p[0]) for p in plans] == [500, 500] and plans[0][3] == plans[1][3] == 1.0
    np.testing.assert_array_equal(np.concatenate([plans[0][0], plans[1][0]]), order[:1000])
    for world, N, B in ((2, 2500, 1000), (4, 65000, 4096), (8, 65000, 4096)):
        order = np.arange(N)
        ps = [epoch_plan(order, B, r, world) for r in range(world)]
        assert len({p[1] for p in ps}) == 1 and ps[0][1] == N // B
        assert all(len(p[0]) == (N // B) * (B // world) and p[2] == 0 and p[3] == 1.0 / (N // B) for p in ps)
        got = np.sort(np.concatenate([p[0] for p in ps]))
        np.testing.assert_array_equal(got, np.arange((N // B) * B))       # every row of the full batches exactly once
    with pytest.raises(ValueError):                                       # N < B: not one full global batch
        epoch_plan(np.arange(65000), 65536, 0, 8)
    with pytest.raises(ValueError):
        epoch_plan(np.arange(100), 30, 0, 4)                              # batch not divisible by the world size
    o, n_full, tail, w = epoch_plan(np.arange(1999), 1000, 0, 1)          # single process: the short last batch stays
    assert (len(o), n_full, tail, w) == (1999, 1, 999, 0.5)


def _epoch_worker(rank, world, port, n_rows, batch, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmvae_hip.parallel import epoch_plan, GradExchange
    ex = GradExchange()
    order = np.random.RandomState(3).permutation(n_rows)
    try:
        mine, n_full, tail, weight = epoch_plan(order, batch, rank, world)
    except ValueError as e:
        out.put((rank, "raised", str(e)))
        dist.destroy_process_group()
        return
    grad = torch.zeros(8)
    for step in range(n_full):          # one gradient exchange per step, exactly as train_op issues them
        grad += float(mine[step * (batch // world)])
        ex(grad)
    loss = ex.mean_scalars([float(n_full)])[0]          # the end-of-epoch scalar exchange
    out.put((rank, n_full, loss))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,batch", [(1999, 1000), (600, 1000)])
def test_epoch_loop_world2_gloo_collectives_pair_up(n_rows, batch):
    """two ranks run the epoch's collective sequence on gloo: N % B > B - world completes with one step each
    (it used to give rank 0 two steps and hang); N < B raises on every rank instead of silently training nothing."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, n_rows, batch, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if n_rows < batch:
        assert [r[1] for r in res] == ["raised", "raised"]
    else:
        assert [(r[1], r[2]) for r in res] == [(1, 1.0), (1, 1.0)]


def _sharded_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), DMVAE_DP_MODE="sharded")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmvae_hip.parallel import make_exchange, ShardedExchange
    ex = make_exchange(param_bytes=0)
    assert isinstance(ex, ShardedExchange) and ex.enabled and ex.sharded and not ex.overlap and ex.align == 64 * world
    n = ex.padded(1000)                                   # an "arena" of 1000 real elements
    assert n % (64 * world) == 0 and n >= 1000
    # plan-style buckets in completion order (decoder+priors, heads, trunk) with unaligned interior bounds
    raw = [(700, 1000), (300, 700), (0, 300)]
    b = ex.bucket_bounds(raw, n)
    assert len(b) == 3 and None not in b
    assert b[0][1] == n and b[-1][0] == 0 and all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))
    assert all((hi - lo) % (64 * world) == 0 for lo, hi in b)
    assert all(bl >= rl for (bl, _), (rl, _) in zip(b, raw))          # a bucket only grows into what was finished EARLIER
    g = torch.Generator().manual_seed(7 + rank)
    grad = torch.randn(n, generator=g)
    ref = grad.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)            # what the all-reduce form would hold everywhere
    param = torch.zeros(n)
    for lo, hi in b:
        ex.reduce_scatter(grad, lo, hi)
        slo, shi = ex.owned(lo, hi)
        assert torch.equal(grad[slo:shi], ref[slo:shi])   # two ranks: a + b either way
        param[slo:shi] = -0.5 * grad[slo:shi]             # the "update" of the owned slice only
        ex.all_gather(param, lo, hi)
    out.put((rank, bool(torch.equal(param, -0.5 * ref)), [tuple(x) for x in b]))
    dist.destroy_process_group()


def test_sharded_exchange_world2_gloo():
    """reduce-scatter -> owned-slice update -> all-gather over padded, rounded buckets reproduces all-reduce +
    replicated update on every rank (SURVEY 8e's preferred collective; the GPU tests run the real Adam on it)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and res[0][2] == res[1][2]


def _self_check_worker(rank, world, port, perturb, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), DMVAE_DP_MODE="sharded")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmvae_hip.parallel import make_exchange
    ex = make_exchange(param_bytes=0)
    n = ex.padded(5000)
    g = torch.Generator().manual_seed(11)                 # the same "weights" on every rank ...
    shadow = torch.randn(n, generator=g).to(torch.bfloat16)
    tail = torch.randn(300, generator=g)
    slo, shi = ex.owned(0, n)
    ex.all_gather(shadow, 0, n)                           # a real in-place gather (every rank takes part) ...
    if perturb == "slice" and rank == 1:                  # ... then ONE rank's copy differs inside a slice (what an aliasing fault of the
        shadow[slo + 7] += 1.0                            # in-place collective, or a lost gather, would leave)
    elif perturb == "tail" and rank == 0:
        tail[5] = -tail[5] if float(tail[5]) != 0.0 else 1.0
    elif perturb == "sign-of-zero":                       # bits, not values: -0.0 == +0.0 as floats
        shadow[3] = 0.0
        if rank == 1:
            shadow[3] = -shadow[3]
    try:
        res = ex.self_check(shadow, n, tail)
    except RuntimeError as e:
        res = "raised: " + str(e)
    out.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("perturb", ["none", "slice", "tail", "sign-of-zero"])
def test_sharded_exchange_self_check_world2_gloo(perturb):
    """the first sharded step of a multi-rank job compares the replicas (checksums of the gathered bf16 shadow and of
    the replicated tail; all-reduce MIN == MAX).  Unperturbed: "ok" on both ranks.  One rank's owned slice (or the tail, or only the
    SIGN of a zero) different: RuntimeError on EVERY rank, naming the all-reduce fallback."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_self_check_worker, args=(r, 2, port, perturb, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if perturb == "none":
        assert [r[1] for r in res] == ["ok", "ok"]
    else:
        assert all(r[1].startswith("raised: ") and "DMVAE_DP_MODE=allreduce" in r[1] for r in res), res
        assert all(("tail" if perturb == "tail" else "weights") in r[1] for r in res), res


def test_png_writer_roundtrip(tmp_path):
    """includes/visualization.py writes its figures with its own PNG encoder: decode it by hand."""
    import struct, zlib
    sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
    from includes import visualization as V
    img = (np.arange(6 * 5).reshape(6, 5) * 8).astype(np.float64)
    path = str(tmp_path / "a" / "t.png")
    V._write_png(path, img)
    blob = open(path, "rb").read()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(blob):
        n, tag = struct.unpack(">I4s", blob[pos:pos + 8])
        data = blob[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", blob[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xffffffff
        chunks[tag] = data
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (5, 6, 8, 0)
    raw = zlib.decompress(chunks[b"IDAT"])
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(6, 6)
    assert (rows[:, 0] == 0).all() and (rows[:, 1:] == img.astype(np.uint8)).all()
    # RGB form + the cluster scatter of mnist_sample_plot(tsne=True) (visualization.py:93-110)
    rng = np.random.RandomState(0)
    sc = V.cluster_scatter([rng.randn(80, 2) + 6 * k for k in range(3)], side=60)
    assert sc.shape == (60, 60, 3) and len({tuple(c) for c in sc.reshape(-1, 3).astype(int).tolist()}) == 4      # white + three cluster colours
    sc5 = V.cluster_scatter([rng.randn(40, 5) + 4 * k for k in range(2)], side=32)                             # > 2 latent dims: through t-SNE
    assert sc5.shape == (32, 32, 3) and (sc5 != 255).any()
    V._write_png(str(tmp_path / "rgb.png"), sc)
    blob = open(str(tmp_path / "rgb.png"), "rb").read()
    assert struct.unpack(">IIBB", blob[16:26]) == (60, 60, 8, 2)
    g = V._grid(np.arange(100 * 4).reshape(100, 4), side=2)
    assert g.shape == (20, 20) and g[0, 0] == 0 and g[0, 2] == 4 and g[2, 0] == 40      # image 1 to the right, image 10 below


# ---- data-parallel layout at the world sizes of the scaling run (VERDICT r2 next #5 ii): the real bucket bounds of the cfg2 /
# cfg4 / cfg5 plans (plan creation is host-only), cut for 4 and 8 ranks, and the in-place reduce-scatter / all-gather on them
_DP_CFGS = {
    "cfg2": dict(input_dim=784, latent_dim=64, n_classes=10, enc=(500, 500), head=2000, dec=(2000, 500, 500), batch=4096),
    "cfg4": dict(input_dim=784, latent_dim=256, n_classes=50, enc=(500, 500), head=2000, dec=(2000, 500, 500), batch=8192),
    "cfg5": dict(input_dim=4096, latent_dim=512, n_classes=256, enc=(4096,) * 4, head=4096, dec=(4096,) * 4, batch=8192),
}


def _plan_buckets(name):
    """(weight elements = where the small-tensor tail begins, [(lo, hi)] of the weight buckets in completion order) of a
    BASELINE config's plan -- no GPU involved"""
    import ctypes as C
    from dmvae_hip import _lib
    k = _DP_CFGS[name]
    cfg = _lib.Config()
    cfg.input_dim, cfg.latent_dim, cfg.n_classes = k["input_dim"], k["latent_dim"], k["n_classes"]
    cfg.n_enc, cfg.head_dim, cfg.n_dec = len(k["enc"]), k["head"], len(k["dec"])
    for i, v in enumerate(k["enc"]):
        cfg.enc[i] = v
    for i, v in enumerate(k["dec"]):
        cfg.dec[i] = v
    cfg.dtype, cfg.max_batch = _lib.BF16, k["batch"]
    cfg.beta1, cfg.beta2, cfg.adam_eps = 0.9, 0.999, 1e-8
    h = C.c_void_p()
    _lib.check(_lib.lib.dmvae_plan_create(C.byref(cfg), C.byref(h)))
    sz = _lib.Sizes()
    _lib.check(_lib.lib.dmvae_plan_sizes(h, C.byref(sz)))
    b = (C.c_int64 * 5)()
    _lib.check(_lib.lib.dmvae_plan_grad_buckets(h, b))
    _lib.lib.dmvae_plan_destroy(h)
    assert b[3] % 4096 == 0 and b[4] == sz.param_elems and 0 < b[4] - b[3] < 0.02 * b[4]      # the tail: every bias + the prior tables, a few per mille
    return int(b[3]), [(b[2], b[3]), (b[1], b[2]), (b[0], b[1])]          # the WEIGHT range (what the sharded exchange cuts) and its buckets


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name", sorted(_DP_CFGS))
def test_sharded_layout_of_the_baseline_plans(name, world):
    """bucket_bounds / owned on the real plans at the world sizes of the scaling run: the padded arena is cut into buckets that
    tile it, every bucket into `world` equal 4-aligned slices, a bucket only grows into what finished earlier, and the slices
    of all ranks tile every bucket exactly once (pure layout arithmetic: no process group)."""
    from dmvae_hip.parallel import ShardedExchange
    n_real, raw = _plan_buckets(name)
    assert raw[0][1] == n_real and raw[-1][0] == 0
    covered = np.zeros(0)
    for rank in range(world):
        ex = ShardedExchange.__new__(ShardedExchange)
        ex.group, ex.enabled, ex.world, ex.rank, ex.align = None, False, world, rank, 64 * world
        n = ex.padded(n_real)
        assert n == n_real                                       # the weight range ends at a multiple of 4096: already aligned for 2 / 4 / 8 ranks
        b = ex.bucket_bounds(raw, n)
        assert len(b) == 3 and None not in b                       # real plans: no bucket collapses
        assert b[0][1] == n and b[-1][0] == 0 and all(b[i][0] == b[i + 1][1] for i in range(2))
        assert all(bl >= rl for (bl, _), (rl, _) in zip(b, raw))
        for lo, hi in b:
            slo, shi = ex.owned(lo, hi)
            assert (hi - lo) % (64 * world) == 0 and (shi - slo) * world == hi - lo and slo % 4 == 0
            assert slo == lo + rank * (shi - slo)
        # the TWO-bucket form (MNIST-sized arenas): decoder + heads as one bucket behind segment 1, the trunk behind segment 2
        raw2 = [(raw[1][0], raw[0][1]), raw[2]]
        b2 = ex.bucket_bounds(raw2, n)
        assert len(b2) == 2 and None not in b2 and b2[0][1] == n and b2[1][0] == 0 and b2[0][0] == b2[1][1] and b2[0][0] >= raw2[0][0]
        for lo, hi in b2:
            slo, shi = ex.owned(lo, hi)
            assert (hi - lo) % (64 * world) == 0 and (shi - slo) * world == hi - lo and slo % 4 == 0
    # a tiny model in a huge world: the middle bucket collapses and must come back as None, in place
    ex = ShardedExchange.__new__(ShardedExchange)
    ex.group, ex.enabled, ex.world, ex.rank, ex.align = None, False, 8, 0, 512
    tiny = ex.bucket_bounds([(700, 1000), (600, 700), (0, 600)], ex.padded(1000))
    assert tiny == [None, None, (0, 1024)]          # both interior cuts round up to the arena's end: everything rides in the LAST segment's bucket
    two = ex.bucket_bounds([(700, 1000), (300, 700), (0, 300)], ex.padded(1000))
    assert two == [None, (512, 1024), (0, 512)]     # (the first segment's elements moved into the second segment's bucket)


def _sharded_worker_n(rank, world, port, cases, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), DMVAE_DP_MODE="sharded")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmvae_hip.parallel import make_exchange
    ok = []
    cases = list(cases) + [(n_real, [(raw[1][0], raw[0][1]), raw[2]]) for n_real, raw in cases[:2]]      # ... and the two-bucket form of the MNIST-sized arenas
    for n_real, raw in cases:
        ex = make_exchange(param_bytes=4 * n_real)
        n = ex.padded(n_real)
        b = [x for x in ex.bucket_bounds(raw, n) if x is not None]
        g = torch.Generator().manual_seed(11 * rank + 3)
        grad = torch.randint(-8, 9, (n,), generator=g).float()        # small integers: every summation order gives the same bits
        ref = grad.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        param = torch.full((n,), float("nan"))
        for lo, hi in b:
            ex.wait(ex.reduce_scatter(grad, lo, hi, async_op=True))
            slo, shi = ex.owned(lo, hi)
            good = torch.equal(grad[slo:shi], ref[slo:shi])
            param[slo:shi] = -0.125 * grad[slo:shi]                   # the owned slice's "update" (1 / world folded in)
            ex.wait(ex.all_gather(param, lo, hi, async_op=True))
            ok.append(good)
        ok.append(bool(torch.equal(param, -0.125 * ref)))
    out.put((rank, all(ok), len(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_exchange_world4_world8_gloo(world):
    """in-place reduce-scatter -> owned-slice update -> in-place all-gather == all-reduce + replicated update on EVERY rank at
    world 4 and 8, on the cfg2- and cfg4-sized arenas with their real bucket bounds and on the cfg5 plan's bounds scaled 1:32
    (its 175 M-element arena x 8 processes does not belong in a CPU test; interior bounds kept unaligned)."""
    import torch.multiprocessing as mp
    cases = [_plan_buckets("cfg2"), _plan_buckets("cfg4")]
    n5, raw5 = _plan_buckets("cfg5")
    sc = lambda v: v // 32 + (7 if 0 < v < n5 else 0)
    cases.append((sc(n5), [(sc(lo), sc(hi)) for lo, hi in raw5]))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker_n, args=(r, world, port, cases, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world)) and all(r[1] for r in res), res


def test_exchange_policy_by_arena_size(monkeypatch):
    """make_exchange: ONE reduce-scatter / all-gather pair after the backward for the MNIST-sized arenas (priced in round 4: the
    bucketed forms cost more than they can hide there), three overlapped buckets from 64 MiB (the 4096-wide stack); the two-bucket
    form (decoder + heads behind segment 1, trunk behind segment 2) on request; the environment switches override each choice."""
    from dmvae_hip.parallel import make_exchange
    for k in ("DMVAE_DP_OVERLAP", "DMVAE_DP_BUCKETS", "DMVAE_DP_MODE"):
        monkeypatch.delenv(k, raising=False)
    mnist, wide = make_exchange(4 * 5955584), make_exchange(4 * 175 * 10 ** 6)
    assert (mnist._overlap, wide._overlap) == (False, True) and wide.n_buckets == 3 and mnist.sharded and wide.sharded
    monkeypatch.setenv("DMVAE_DP_OVERLAP", "1")
    monkeypatch.setenv("DMVAE_DP_BUCKETS", "2")
    two = make_exchange(4 * 5955584)
    assert two._overlap is True and two.n_buckets == 2
    monkeypatch.setenv("DMVAE_DP_OVERLAP", "0")
    assert make_exchange(4 * 175 * 10 ** 6)._overlap is False
    monkeypatch.setenv("DMVAE_DP_MODE", "allreduce")
    assert not make_exchange(4 * 5955584).sharded


def test_asm_register_loads_of_the_streaming_kernel_are_covered_by_waits(tmp_path):
    """csrc/heads_dx.hip loads its ReLU gates from inline asm (so that its counted vmcnt waits are exact).  hipcc regards such a
    destination as written at the asm statement and may copy it while the load is in flight -- the first form of the kernel kept the
    gates in registers across a loop join, the compiler's v_mov_b64 there read them early, and a few per cent of some tiles' masks came
    out wrong on some launches (round 4).  tools/audit_asm_loads.py checks the GENERATED ISA: before anything touches such a register
    there is a covering counted wait, in a branch-free stretch.  Here: the auditor flags a synthetic uncovered copy, passes the same
    code with the wait in place, and passes the kernel as built."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(ROOT, "tools", "audit_asm_loads.py"))
    aud = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aud)
    head = "toy_kernel:\n\tglobal_load_dwordx2 v[4:5], v[0:1], off\n\tglobal_load_lds_dwordx4 v2, s[0:1]\n"
    tail = "\tv_mov_b64_e32 v[8:9], v[4:5]\n\ts_endpgm\n.Lfunc_end0:\n"
    for name, mid, want in (("uncovered", "\ts_waitcnt vmcnt(2)\n", 1), ("covered", "\ts_waitcnt vmcnt(1)\n", 0),
                            ("join", "\ts_waitcnt vmcnt(1)\n.LBB0_1:\n", 1)):
        f = tmp_path / (name + ".s")
        f.write_text(head + mid + tail)
        assert aud.audit(str(f), "toy_kernel") == want, name
    out = tmp_path / "heads_dx.s"
    src = os.path.join(ROOT, "deep-mixture-vae_amd", "csrc", "heads_dx.hip")
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                        "-S", "--cuda-device-only", src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert aud.audit(str(out), "heads_dx_stream_kernel") == 0
    txt = out.read_text()
    assert ".vgpr_spill_count: 0" in txt and "scratch_" not in txt.split(".amdgpu_metadata")[0]
