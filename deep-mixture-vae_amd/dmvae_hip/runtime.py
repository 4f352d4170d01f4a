"""Host runtime of the MI355X DMVAE step: Session (the stand-in for tf.Session,
code/train.py:238) and StepEngine (the step plan of libdmvae_hip.so bound to
torch-owned device memory).  PyTorch is used for device memory, streams, HIP
graphs and torch.distributed only; all arithmetic runs in the HIP library."""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "dmvae_hip needs an AMD GPU (gfx950): torch.cuda.is_available() is False. "
            "There is no CPU execution path in this package.")


class Session:
    """Opaque runtime handle: device, stream, rank/world.  Mirrors the role of
    the tf.Session the reference passes around (train.py:238, base_models.py:112)."""

    def __init__(self, device=None):
        _require_gpu()
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if device is None:
            device = torch.device("cuda", self.local_rank % max(1, torch.cuda.device_count()))
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device)

    def run(self, *a, **k):   # the reference's session.run has no counterpart: fail loudly
        raise NotImplementedError("Session.run: there is no TensorFlow graph; use the model methods")

    def close(self):
        pass


_default_session = None


def default_session():
    global _default_session
    if _default_session is None:
        _default_session = Session()
    return _default_session


# the conv stack of the checked-in encoder, base_models.py:181-201: (name, cin, cout, image side, pooled)
CONV_STACK = (("conv0", 1, 32, 28, False), ("conv1", 32, 32, 28, True), ("conv2", 32, 64, 14, False),
              ("conv3", 64, 64, 14, True), ("conv4", 64, 128, 7, False), ("conv5", 128, 128, 7, True))
CONV_FLAT = 4 * 4 * 128


def vade_layer_table(input_dim, latent_dim, enc_layers, dec_layers, cnn=False):
    """VaDE (base_models.py:490-547): FullyConnected encoder / decoder layers (xavier bias), tf.layers.dense mean /
    log_var straight off the trunk and output layer (zero bias).  cnn (:456-488): the encoder's dense part is the one
    ("fc", 2048 -> 128) layer behind the conv stack."""
    t, prev = [], (CONV_FLAT if cnn else input_dim)
    for i, h in enumerate(enc_layers):
        t.append(("enc%d" % i, prev, h, "xavier"))
        prev = h
    t += [("mean", prev, latent_dim, "zero"), ("logvar", prev, latent_dim, "zero")]
    prev = latent_dim
    for i, h in enumerate(dec_layers):
        t.append(("dec%d" % i, prev, h, "xavier"))
        prev = h
    t.append(("out", prev, input_dim, "zero"))
    return t


def layer_table(input_dim, latent_dim, n_classes, enc_layers, head_dim, dec_layers, cnn=False):
    """[(name, fan_in, fan_out, bias_kind)] in graph-construction order
    (base_models.py:218-293).  bias_kind: 'zero' = tf.layers.dense default,
    'xavier' = FullyConnected bias (1,out) (includes/layers.py:24-28).  cnn: the
    trunk's dense part is the one "fc" layer 2048 -> enc_layers[0] of base_models.py:202."""
    t = []
    prev = CONV_FLAT if cnn else input_dim
    for i, h in enumerate(enc_layers):
        t.append(("enc%d" % i, prev, h, "xavier" if cnn else "zero"))
        prev = h
    trunk = prev
    t += [("zh", trunk, head_dim, "zero"), ("mean", head_dim, latent_dim, "zero"),
          ("logvar", head_dim, latent_dim, "zero"), ("ch", trunk, head_dim, "zero"),
          ("logits", head_dim, n_classes, "zero")]
    prev = latent_dim
    for i, h in enumerate(dec_layers):
        t.append(("dec%d" % i, prev, h, "xavier"))
        prev = h
    t.append(("out", prev, input_dim, "zero"))
    return t


PIPELINE_MAX_BATCH = 2048      # capture_step: batches up to this many rows assemble the next batch under the running step by default (measured: see capture_step)


class StepEngine:
    """One DMVAE model instance on one GPU: parameter / gradient / Adam arenas,
    activation workspace, device step state, and the enqueue methods of the
    step plan.  All compute calls go through the C ABI."""

    def __init__(self, input_dim, latent_dim, n_classes, enc_layers=(500, 500), head_dim=2000,
                 dec_layers=(2000, 500, 500), input_type="binary", dtype="bf16", max_batch=100,
                 mode="exact", temperature=1.0, seed=0, deterministic=False, session=None,
                 beta1=0.9, beta2=0.999, adam_eps=1e-8, cnn=False, model="dmvae", adam_ieee=False):
        self.cnn = bool(cnn)
        self.model = model
        if model not in ("dmvae", "vade"):
            raise ValueError("model must be 'dmvae' or 'vade'")
        self.session = session or default_session()
        dev = self.session.device
        self.device = dev
        self.input_dim, self.latent_dim, self.n_classes = int(input_dim), int(latent_dim), int(n_classes)
        self.enc_layers, self.dec_layers, self.head_dim = tuple(enc_layers), tuple(dec_layers), int(head_dim)
        self.input_type = input_type
        self.dtype = {"bf16": _lib.BF16, "fp32": _lib.F32, "f32": _lib.F32}[dtype]
        self.dtype_name = "bf16" if self.dtype == _lib.BF16 else "fp32"
        self.max_batch = int(max_batch)
        self.mode = {"exact": 0, "relaxed": 1}[mode]
        self.deterministic = bool(deterministic)
        cfg = _lib.Config()
        cfg.input_dim, cfg.latent_dim, cfg.n_classes = self.input_dim, self.latent_dim, self.n_classes
        cfg.n_enc = len(self.enc_layers)
        for i, v in enumerate(self.enc_layers):
            cfg.enc[i] = int(v)
        cfg.head_dim = self.head_dim
        cfg.n_dec = len(self.dec_layers)
        for i, v in enumerate(self.dec_layers):
            cfg.dec[i] = int(v)
        cfg.input_type = {"binary": 0, "real": 1}[input_type]
        cfg.dtype = self.dtype
        cfg.max_batch = self.max_batch
        cfg.mode = self.mode
        cfg.temperature = float(temperature)
        cfg.beta1, cfg.beta2, cfg.adam_eps = float(beta1), float(beta2), float(adam_eps)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.deterministic = 1 if deterministic else 0
        cfg.trunk = _lib.TRUNK_CNN if self.cnn else _lib.TRUNK_MLP
        cfg.model = _lib.MODEL_VADE if model == "vade" else _lib.MODEL_DMVAE
        # bf16 plans: TF-Adam's quotient with the IEEE square root and division (the fp32 mode's arithmetic) instead of the
        # hardware v_sqrt_f32 / v_rcp_f32 (1 ulp each, the default)
        self.adam_ieee = bool(adam_ieee)
        cfg.adam_ieee = 1 if adam_ieee else 0
        self._cfg = cfg
        h = C.c_void_p()
        check(lib.dmvae_plan_create(C.byref(cfg), C.byref(h)), "dmvae_plan_create")
        self._plan = h
        sz = _lib.Sizes()
        check(lib.dmvae_plan_sizes(self._plan, C.byref(sz)), "dmvae_plan_sizes")
        self.sizes = sz
        self.batch_pad, self.input_pad = sz.batch_pad, sz.input_pad
        # arenas padded to a multiple of 64 * world elements: the sharded exchange (parallel.ShardedExchange) cuts
        # them into world equal slices; the pad stays zero (zero gradient -> zero Adam update)
        gran = 64 * max(1, self.session.world_size)
        n = (sz.param_elems + gran - 1) // gran * gran
        self.param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.param_bf16 = torch.zeros(n, dtype=torch.bfloat16, device=dev) if self.dtype == _lib.BF16 else None
        self.work = torch.zeros(sz.work_bytes, dtype=torch.uint8, device=dev)
        self.state_t = torch.zeros(C.sizeof(_lib.State), dtype=torch.uint8, device=dev)
        b = _lib.Buffers(ptr(self.param), ptr(self.grad), ptr(self.m), ptr(self.v),
                         ptr(self.param_bf16), ptr(self.work), ptr(self.state_t), n)
        check(lib.dmvae_plan_bind(self._plan, C.byref(b)), "dmvae_plan_bind")
        self.tensors = {}
        for i in range(sz.n_tensors):
            ti = _lib.TensorInfo()
            check(lib.dmvae_plan_tensor(self._plan, i, C.byref(ti)), "dmvae_plan_tensor")
            self.tensors[ti.name.decode()] = (ti.offset, ti.rows, ti.cols, ti.ld)
        self.write_state(kl_ratio=1.0, lr=0.002, epoch_weight=1.0, batches_per_epoch=0)
        self._graph = None
        self._pf_primed = False        # capture_step's pipelined form: the current batch buffer holds the batch of the device cursor
        self._xsel = 0

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                lib.dmvae_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ------------------------------------------------------------ parameters
    def _strided(self, arena, name):
        off, rows, cols, ld = self.tensors[name]
        if name.startswith("b_"):
            return torch.as_strided(arena, (cols,), (1,), off)
        return torch.as_strided(arena, (rows, cols), (ld, 1), off)

    def param_view(self, name):
        return self._strided(self.param, name)

    def grad_view(self, name):
        return self._strided(self.grad, name)

    def parameter_names(self):
        return list(self.tensors.keys())

    def _to_host(self, view, name):
        return view.detach().cpu().numpy().copy()

    def _require_current_master(self, what):
        """Sharded bf16 data-parallel steps (ShardedExchange) keep a rank's fp32 master weights -- and its Adam moments --
        current on its OWNED slice only; the rest of the fp32 arena is stale until sync_master() (a collective) has run.
        Everything that reads or updates the whole fp32 arena refuses to run on stale data instead of returning it."""
        if getattr(self, "_master_stale", False):
            raise RuntimeError("%s: the fp32 master weights are stale outside this rank's slice after sharded data-parallel steps; "
                               "call StepEngine.sync_master(exchange) on EVERY rank first (it is a collective)" % what)

    def get_parameters(self):
        self._require_current_master("get_parameters")
        return {k: self._to_host(self.param_view(k), k) for k in self.tensors}

    def get_gradients(self):
        return {k: self._to_host(self.grad_view(k), k) for k in self.tensors}

    def set_parameters(self, params):
        if set(self.tensors) <= set(params):          # every tensor is overwritten: nothing stale survives
            self._master_stale = False
        self._require_current_master("set_parameters (it re-derives the whole bf16 shadow from the fp32 arena)")
        for k, v in params.items():
            dst = self.param_view(k)
            dst.copy_(torch.as_tensor(np.asarray(v, dtype=np.float32)).to(self.device).reshape(dst.shape))
        self.refresh_shadow()

    def init_parameters(self, seed=0):
        """Reference-faithful initialisation: xavier-uniform
        kernels (train.py:197), zero dense biases, xavier FullyConnected biases
        (includes/layers.py:24-28), prior means ~ N(0,1), prior log-vars = 0
        (priors.py:57-65)."""
        rng = np.random.RandomState(seed)
        p = {}
        if self.cnn:
            # Convolution (includes/layers.py:39-51): kernel (3,3,cin,cout) and bias (cout,) both xavier;
            # TF's fans: 9*cin / 9*cout for the kernel, n / n for a 1-D shape.  Stored HWIO-flattened.
            for name, ci, co, _, _ in CONV_STACK:
                lim = math.sqrt(6.0 / (9 * ci + 9 * co))
                p["W_" + name] = rng.uniform(-lim, lim, size=(9 * ci, co))
                lb = math.sqrt(6.0 / (co + co))
                p["b_" + name] = rng.uniform(-lb, lb, size=(co,))
        table = (vade_layer_table(self.input_dim, self.latent_dim, self.enc_layers, self.dec_layers, self.cnn) if self.model == "vade" else
                 layer_table(self.input_dim, self.latent_dim, self.n_classes, self.enc_layers, self.head_dim, self.dec_layers, self.cnn))
        for name, fi, fo, bk in table:
            lim = math.sqrt(6.0 / (fi + fo))
            p["W_" + name] = rng.uniform(-lim, lim, size=(fi, fo))
            if bk == "zero":
                p["b_" + name] = np.zeros((fo,))
            else:
                lb = math.sqrt(6.0 / (1 + fo))
                p["b_" + name] = rng.uniform(-lb, lb, size=(fo,))
        p["prior_means"] = rng.randn(self.n_classes, self.latent_dim)
        p["prior_log_vars"] = np.zeros((self.n_classes, self.latent_dim))
        self.param.zero_()
        self.m.zero_()
        self.v.zero_()
        self._moments_sharded = False
        self.grad.zero_()
        self.set_parameters(p)
        st = self.read_state()
        self.write_state(adam_t=0, noise_step=0, kl_ratio=st.kl_ratio, lr=st.lr)

    def refresh_shadow(self):
        self._require_current_master("refresh_shadow")
        if self.param_bf16 is not None:
            check(lib.dmvae_cast_f32_to_bf16(self._stream(), ptr(self.param), ptr(self.param_bf16),
                                             self.param.numel()), "dmvae_cast_f32_to_bf16")

    # ------------------------------------------------------------ device step state
    def read_state(self):
        raw = self.state_t.cpu().numpy().tobytes()
        return _lib.State.from_buffer_copy(raw)

    def write_state(self, **kw):
        st = self.read_state() if kw.pop("_merge", True) else _lib.State()
        for k, v in kw.items():
            setattr(st, k, v)
        host = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8)
        self.state_t.copy_(host)
        self._pf_primed = False        # (the batch cursor may have moved)

    def reset_optimizer(self, lr=None):
        """A fresh tf.train.AdamOptimizer instance (base_models.py:102, :307-320 create one per
        training stage): zero m and v, t back to 0, optionally a new learning rate."""
        self.m.zero_()
        self.v.zero_()
        self._moments_sharded = False
        kw = dict(adam_t=0, lr_t=0.0)
        if lr is not None:
            kw["lr"] = float(lr)
        self.write_state(**kw)

    def reset_epoch(self, batches_per_epoch, kl_ratio=None, epoch_weight=None):
        kw = dict(batch_cursor=0, batches_per_epoch=int(batches_per_epoch), epoch_loss=0.0, epoch_recon=0.0,
                  epoch_klz=0.0, epoch_klc=0.0,
                  epoch_weight=float(epoch_weight if epoch_weight is not None else 1.0 / max(1, batches_per_epoch)))
        if kl_ratio is not None:
            kw["kl_ratio"] = float(kl_ratio)
        self.write_state(**kw)

    # ------------------------------------------------------------ enqueue
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_batch(self, data, perm=None, first=0, n_valid=None, use_state_cursor=False):
        """Dataset.get_batches batch assembly (includes/utils.py:449-463) on the
        device: data f32 [N, input_dim] resident in HBM, perm int32 [N] or None."""
        assert data.dtype == torch.float32 and data.is_contiguous() and data.shape[1] == self.input_dim
        assert perm is None or (perm.dtype == torch.int32 and perm.is_contiguous())
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        self._pf_primed = False
        check(lib.dmvae_plan_load_batch(self._plan, self._stream(), ptr(data), data.shape[0], ptr(perm),
                                        int(first), n_valid, 1 if use_state_cursor else 0), "dmvae_plan_load_batch")

    def _load_batch_for_step(self, data, perm=None, first=0, n_valid=None, use_state_cursor=False):
        """load_batch for the step paths (ONE forward_backward / train step follows at once): dmvae_plan_load_batch_step -- on bf16
        plans only the bf16 copy of the batch is written and the reconstruction epilogue reads its targets from `data` through
        `perm` (both must stay alive until the step has run; the "x" view is not valid afterwards)."""
        assert data.dtype == torch.float32 and data.is_contiguous() and data.shape[1] == self.input_dim
        assert perm is None or (perm.dtype == torch.int32 and perm.is_contiguous())
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        self._step_src = (data, perm)                       # keep them alive
        self._pf_primed = False
        check(lib.dmvae_plan_load_batch_step(self._plan, self._stream(), ptr(data), data.shape[0], ptr(perm),
                                             int(first), n_valid, 1 if use_state_cursor else 0), "dmvae_plan_load_batch_step")

    def forward_backward(self, n_valid=None, eps=None, gumbel=None, inv_B=None):
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        if eps is not None:
            assert eps.dtype == torch.float32 and eps.is_contiguous() and eps.shape == (n_valid, self.latent_dim)
        if gumbel is not None:
            assert gumbel.dtype == torch.float32 and gumbel.is_contiguous() and gumbel.shape == (n_valid, self.n_classes)
        inv_B = 1.0 / n_valid if inv_B is None else float(inv_B)
        self._pf_primed = False            # (moves the device cursor: the pipelined callable must re-assemble its batch)
        check(lib.dmvae_plan_forward_backward(self._plan, self._stream(), n_valid, ptr(eps), self.latent_dim,
                                              ptr(gumbel), self.n_classes, inv_B), "dmvae_plan_forward_backward")

    def forward_backward_stage(self, stage, n_valid=None, eps=None, gumbel=None, inv_B=None):
        """segment `stage` (0, 1, 2) of forward_backward; after it grad_buckets()[stage] is final."""
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        inv_B = 1.0 / n_valid if inv_B is None else float(inv_B)
        check(lib.dmvae_plan_forward_backward_stage(self._plan, self._stream(), int(stage), n_valid, ptr(eps), self.latent_dim,
                                                    ptr(gumbel), self.n_classes, inv_B), "dmvae_plan_forward_backward_stage")

    def grad_buckets(self):
        """([(lo, hi)] * 3, (tail_lo, tail_hi)): the element ranges of the WEIGHT part of the gradient arena in the order the
        three segments complete them (decoder, heads, trunk), and the tail -- every bias, then the prior tables -- which is
        complete when the last segment has run (dmvae_plan_grad_buckets)."""
        b = (C.c_int64 * 5)()
        check(lib.dmvae_plan_grad_buckets(self._plan, b), "dmvae_plan_grad_buckets")
        return [(b[2], b[3]), (b[1], b[2]), (b[0], b[1])], (b[3], b[4])

    def set_stage_groups(self, n):
        """2: segments 0 and 1 of the staged backward share ONE weight-gradient launch (decoder + heads complete after segment 1);
        3: one launch per segment (dmvae_plan_set_stage_groups)."""
        check(lib.dmvae_plan_set_stage_groups(self._plan, int(n)), "dmvae_plan_set_stage_groups")

    def update_range(self, lo, hi, grad_scale=1.0):
        check(lib.dmvae_plan_update_range(self._plan, self._stream(), float(grad_scale), int(lo), int(hi)), "dmvae_plan_update_range")

    def _stage_groups(self, grad_sync, buckets):
        """The overlapped exchange's grouping of the three backward segments: 3 buckets = a collective behind every segment; 2 buckets
        (grad_sync.n_buckets, parallel.make_exchange) = decoder + heads behind segment 1 as ONE bucket (one weight-gradient launch
        for both, dmvae_plan_set_stage_groups), the trunk behind segment 2.  Returns (segments per group, weight bucket per group)."""
        nb = int(getattr(grad_sync, "n_buckets", 3) or 3)
        if getattr(self, "_stage_groups_set", None) != nb:
            self.set_stage_groups(2 if nb == 2 else 3)
            self._stage_groups_set = nb
        if nb == 2:
            (lo0, hi0), (lo1, hi1), trunk = buckets
            assert hi1 == lo0          # heads lie right below the decoder in the arena
            return [(0, 1), (2,)], [(lo1, hi0), trunk]
        return [(0,), (1,), (2,)], list(buckets)

    def _step_with_exchange(self, grad_sync, grad_scale, n_valid=None, eps=None, gumbel=None, inv_B=None):
        """forward + backward + gradient exchange + Adam.  Bucketed and overlapped when the exchange says so
        (`overlap`): each weight bucket's collective starts right behind the backward segment that completes it and its
        Adam runs as soon as its sum has landed, while later buckets are still in flight; else one collective per range
        after the whole backward.

        Sharded exchange (parallel.ShardedExchange): the WEIGHT range is reduce-scattered, TF-Adam runs on the owned slice
        only (m and v are never touched elsewhere) and the updated weights are all-gathered -- on bf16 plans as their bf16
        SHADOW, 2 B per parameter instead of 4 (the GEMMs read nothing else of a weight); a rank's fp32 copy of the weights
        it does not own is then STALE until sync_master() (checkpoints, get_parameters).  The tail (every bias, the prior
        tables: read in fp32 by the epilogues and the latent kernel, 0.3 % of the arena) is all-reduced whole and updated on
        every rank, so the replicas agree in every bit the step reads.  Else: all-reduce + replicated Adam."""
        buckets, (tlo, thi) = self.grad_buckets()
        ev = getattr(self, "_exch_events", None)        # measure_exchange: (backward's last kernel enqueued, step's last work enqueued)
        mark = (lambda: ev[0].record(torch.cuda.current_stream(self.device))) if ev else (lambda: None)
        if getattr(grad_sync, "sharded", False):
            gather = self.param_bf16 if self.param_bf16 is not None else self.param      # what the all-gather carries
            # the sharded part ends at a multiple of 64 * world (tail_off is a multiple of 4096: the same thing for every world
            # that divides 64); whatever lies between is exchanged with the tail
            tlo = tlo // grad_sync.align * grad_sync.align
            if grad_sync.overlap:
                groups, buckets = self._stage_groups(grad_sync, buckets)
                wb = grad_sync.bucket_bounds(buckets, tlo)
                assert len(wb) == len(groups)          # one per group of segments; None = emptied by the rounding (its elements ride in a later bucket)
                handles = []
                for gi, stages in enumerate(groups):
                    for stage in stages:
                        self.forward_backward_stage(stage, n_valid, eps, gumbel, inv_B)
                    if gi + 1 == len(groups):
                        mark()
                    if wb[gi] is not None:
                        lo, hi = wb[gi]
                        handles.append(grad_sync.reduce_scatter(self.grad, lo, hi, async_op=True))
                th = grad_sync.start(self.grad[tlo:thi])
                wb = [b for b in wb if b is not None]
                gathers = []
                for h, (lo, hi) in zip(handles, wb):
                    grad_sync.wait(h)
                    slo, shi = grad_sync.owned(lo, hi)
                    self.update_range(slo, shi, grad_scale)
                    gathers.append(grad_sync.all_gather(gather, lo, hi, async_op=True))
                grad_sync.wait(th)
                self.update_range(tlo, thi, grad_scale)
                grad_sync.finish(gathers)
                self._shard_ranges = wb
            else:
                self.forward_backward(n_valid, eps, gumbel, inv_B)
                mark()
                grad_sync.reduce_scatter(self.grad, 0, tlo)
                grad_sync(self.grad[tlo:thi])
                slo, shi = grad_sync.owned(0, tlo)
                self.update_range(slo, shi, grad_scale)
                self.update_range(tlo, thi, grad_scale)
                grad_sync.all_gather(gather, 0, tlo)
                self._shard_ranges = [(0, tlo)]
            self._master_stale = self.param_bf16 is not None and grad_sync.world > 1
            self._moments_sharded = grad_sync.world > 1      # m, v are current on the owned slice only until reset_optimizer
            if grad_sync.world > 1 and getattr(self, "_exchange_check", None) is None:
                # the FIRST sharded step of this engine on a real multi-rank group: every bit a step reads must be identical on
                # all ranks (parallel.ShardedExchange.self_check: raises on every rank, naming DMVAE_DP_MODE=allreduce)
                torch.cuda.synchronize(self.device)
                self._exchange_check = grad_sync.self_check(gather, tlo, self.param[tlo:thi])
            return
        if getattr(grad_sync, "overlap", False):
            groups, buckets = self._stage_groups(grad_sync, buckets)
            handles = []
            for gi, (stages, (lo, hi)) in enumerate(zip(groups, buckets)):
                for stage in stages:
                    self.forward_backward_stage(stage, n_valid, eps, gumbel, inv_B)
                if gi + 1 == len(groups):
                    mark()
                handles.append(grad_sync.start(self.grad[lo:hi]))
            handles.append(grad_sync.start(self.grad[tlo:thi]))
            for h, (lo, hi) in zip(handles, buckets + [(tlo, thi)]):
                grad_sync.wait(h)
                self.update_range(lo, hi, grad_scale)
        else:
            self.forward_backward(n_valid, eps, gumbel, inv_B)
            mark()
            grad_sync(self.grad)
            self.update(grad_scale)

    def measure_exchange(self, data, perm, grad_sync, grad_scale, steps=10):
        """What of a data-parallel step lies BEHIND its backward pass: `steps` eager steps, each with an event recorded on the compute
        stream right behind the backward pass's last kernel and one behind the step's last enqueue (by then the compute stream has
        waited for every collective).  exposed_us = that tail: the collectives that no backward segment covers, plus the Adam
        launches; step_us = the whole step (event pair around it).  A measurement aid for bench.py --gpus N (the first multi-rank
        run should explain itself); the training state advances by `steps` steps."""
        s = torch.cuda.current_stream(self.device)
        e_b = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        e_0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        e_1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        for i in range(steps):
            e_0[i].record(s)
            self._exch_events = (e_b[i],)
            try:
                self.train_step(data, perm, None, None, None, 0, True, grad_sync, grad_scale)
            finally:
                self._exch_events = None
            e_1[i].record(s)
        torch.cuda.synchronize(self.device)
        tail = sorted(e_b[i].elapsed_time(e_1[i]) for i in range(steps))
        whole = sorted(e_0[i].elapsed_time(e_1[i]) for i in range(steps))
        return {"exposed_us": round(1e3 * tail[steps // 2], 1), "step_us": round(1e3 * whole[steps // 2], 1), "steps": steps,
                "what": "median over eager steps: exposed_us = end of the backward pass's last kernel -> end of the step (collectives not covered "
                        "by a backward segment + Adam launches); step_us = the whole eager step"}

    def sync_master(self, grad_sync):
        """COLLECTIVE (every rank calls it): after sharded bf16 steps a rank's fp32 weights are current on its owned slice only;
        this all-gathers the fp32 master weights (in place) so that get_parameters / a checkpoint see the trained model.  A
        no-op when nothing is stale.  get_parameters / state_dict / update / forward_backward_update RAISE while the master is
        stale (_require_current_master).  The Adam moments m, v stay sharded: a rank only ever holds its own slice's, so
        checkpoints are WEIGHTS ONLY (as the reference's Saver(TRAINABLE_VARIABLES), train.py:233-236) and a resumed job
        starts a fresh optimizer -- also under another world size or exchange mode."""
        if getattr(self, "_master_stale", False) and getattr(grad_sync, "sharded", False):
            for lo, hi in self._shard_ranges:          # the ranges the steps cut into owned slices (one per bucket when overlapped)
                grad_sync.all_gather(self.param, lo, hi)
            torch.cuda.synchronize(self.device)
        self._master_stale = False

    def _require_whole_moments(self, what):
        """sync_master all-gathers the fp32 WEIGHTS; the Adam moments m, v stay current on each rank's own slice only.  A
        replicated update over the whole arena with such moments would let the replicas drift apart silently: refuse until
        reset_optimizer (a fresh optimizer: what every product path does between stages)."""
        if getattr(self, "_moments_sharded", False):
            raise RuntimeError("%s: the Adam moments m, v are current on this rank's slice only after sharded data-parallel steps "
                               "(sync_master gathers the weights, not the moments); call reset_optimizer() first" % what)

    def update(self, grad_scale=1.0):
        self._require_current_master("update (replicated Adam over the whole arena)")
        self._require_whole_moments("update (replicated Adam over the whole arena)")
        check(lib.dmvae_plan_update(self._plan, self._stream(), float(grad_scale)), "dmvae_plan_update")

    def forward_backward_update(self, n_valid=None, eps=None, gumbel=None, inv_B=None):
        """forward + loss + backward + Adam with the update fused into the dW launch
        (dmvae_plan_train_step): single-process training, the gradient arena is not written."""
        self._require_current_master("forward_backward_update (fused Adam over the whole arena)")
        self._require_whole_moments("forward_backward_update (fused Adam over the whole arena)")
        self._pf_primed = False            # a step outside the pipelined callable's replays consumes the batch and moves the cursor
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        if eps is not None:
            assert eps.dtype == torch.float32 and eps.is_contiguous() and eps.shape == (n_valid, self.latent_dim)
        if gumbel is not None:
            assert gumbel.dtype == torch.float32 and gumbel.is_contiguous() and gumbel.shape == (n_valid, self.n_classes)
        inv_B = 1.0 / n_valid if inv_B is None else float(inv_B)
        check(lib.dmvae_plan_train_step(self._plan, self._stream(), n_valid, ptr(eps), self.latent_dim,
                                        ptr(gumbel), self.n_classes, inv_B), "dmvae_plan_train_step")

    def encode(self, n_valid=None):
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        check(lib.dmvae_plan_encode(self._plan, self._stream(), n_valid), "dmvae_plan_encode")

    def decode(self, Z):
        assert Z.dtype == torch.float32 and Z.is_contiguous() and Z.shape[1] == self.latent_dim
        check(lib.dmvae_plan_decode(self._plan, self._stream(), ptr(Z), self.latent_dim, Z.shape[0]), "dmvae_plan_decode")

    def view(self, name, rows=None, cols=None):
        """torch view of a workspace tensor ("mean", "log_var", "logits", "weights",
        "recon", "x", "Z", "dxlogits")."""
        p, ld, dt = C.c_void_p(), C.c_int64(), C.c_int32()
        check(lib.dmvae_plan_view(self._plan, name.encode(), C.byref(p), C.byref(ld), C.byref(dt)), "dmvae_plan_view")
        tdt = torch.bfloat16 if dt.value == _lib.BF16 else torch.float32
        es = 2 if dt.value == _lib.BF16 else 4
        off = p.value - self.work.data_ptr()
        if name.startswith("conv"):          # zero-bordered [batch_pad][hw+2][hw+2][channels] relu outputs of a conv layer
            hw = CONV_STACK[int(name[4])][3]
            P = hw + 2
            flat = self.work[off: off + self.batch_pad * P * P * ld.value * es].view(tdt)
            t = torch.as_strided(flat, (self.batch_pad, P, P, ld.value), (P * P * ld.value, P * ld.value, ld.value, 1))
            nb = self.max_batch if rows is None else rows
            return t[:nb, 1:-1, 1:-1, : (CONV_STACK[int(name[4])][2] if cols is None else cols)]
        flat = self.work[off: off + self.batch_pad * ld.value * es].view(tdt)
        t = torch.as_strided(flat, (self.batch_pad, ld.value), (ld.value, 1))
        full_cols = {"mean": self.latent_dim, "log_var": self.latent_dim, "logits": self.n_classes,
                     "weights": self.n_classes, "recon": self.input_dim, "x": self.input_dim,
                     "Z": self.latent_dim, "dxlogits": self.input_dim}.get(name, ld.value)
        return t[: (self.max_batch if rows is None else rows), : (full_cols if cols is None else cols)]

    def hidden_activations(self, rows=None):
        """{layer name: activation [rows, width]} of the ReLU layers of the last
        forward pass, under the names the parameters use (enc<i>, zh, ch, dec<i>)."""
        out = {}
        if self.model == "vade":
            if self.cnn:
                for name, _, _, _, _ in CONV_STACK:
                    out[name] = self.view(name, rows)
            for i, w in enumerate(self.enc_layers):
                out["enc%d" % i] = self.view("enc%d" % i, rows, w)
            for i, w in enumerate(self.dec_layers):
                out["dec%d" % i] = self.view("dec%d" % i, rows, w)
            return out
        hp = self.view("hzc", rows).shape[1] // 2
        if self.cnn:                          # [rows, H, W, channels] relu outputs of the conv layers
            for i, (name, _, co, hw, _) in enumerate(CONV_STACK):
                out[name] = self.view(name, rows)
        for i, w in enumerate(self.enc_layers):
            out["enc%d" % i] = self.view("enc%d" % i, rows, w)
        hzc = self.view("hzc", rows)
        out["zh"] = hzc[:, : self.head_dim]
        out["ch"] = hzc[:, hp: hp + self.head_dim]
        for i, w in enumerate(self.dec_layers):
            out["dec%d" % i] = self.view("dec%d" % i, rows, w)
        return out

    # ------------------------------------------------------------ whole step, HIP graph
    def train_step(self, data, perm, n_valid=None, eps=None, gumbel=None, first=0, use_state_cursor=False,
                   grad_sync=None, grad_scale=1.0, inv_B=None, fused=None):
        """load batch -> forward/loss/backward -> (gradient exchange) -> Adam.
        fused (default: whenever there is no gradient exchange and grad_scale == 1): the update
        rides in the epilogue of the dW launch; the gradient arena is then not written."""
        self._load_batch_for_step(data, perm, first, n_valid, use_state_cursor)
        if fused is None:
            fused = grad_sync is None and float(grad_scale) == 1.0
        if fused:
            assert grad_sync is None and float(grad_scale) == 1.0
            self.forward_backward_update(n_valid, eps, gumbel, inv_B)
            return
        if grad_sync is not None:
            self._step_with_exchange(grad_sync, grad_scale, n_valid, eps, gumbel, inv_B)
            return
        self.forward_backward(n_valid, eps, gumbel, inv_B)
        self.update(grad_scale)

    def _prefetch_batch(self, data, perm, first=0, n_valid=None, use_state_cursor=True):
        """dmvae_plan_prefetch_batch: the next forward + backward pass also assembles the NEXT batch (host state only: nothing is enqueued)."""
        n_valid = self.max_batch if n_valid is None else int(n_valid)
        self._step_src = (data, perm)
        check(lib.dmvae_plan_prefetch_batch(self._plan, ptr(data), data.shape[0], ptr(perm), int(first), n_valid, 1 if use_state_cursor else 0),
              "dmvae_plan_prefetch_batch")

    def _prefetch_ok(self, data, perm, side):
        """assembles the batch of the device cursor (eagerly) and asks the plan whether it can prefetch behind it"""
        if self.dtype != _lib.BF16:
            return False
        with torch.cuda.stream(side):
            self._load_batch_for_step(data, perm, 0, None, True)
        side.synchronize()
        rc = lib.dmvae_plan_prefetch_batch(self._plan, ptr(data), data.shape[0], ptr(perm), 0, self.max_batch, 1)
        if rc == _lib.EUNSUPPORTED:
            return False
        check(rc, "dmvae_plan_prefetch_batch")
        return True

    def capture_step(self, data, perm, grad_sync=None, grad_scale=1.0, inv_B=None, pipelined=None):
        """Capture one full-batch step (device Philox noise, batch cursor read from
        the device state) into a HIP graph; returns a callable that replays it.
        pipelined: the step no longer starts with its batch gather -- each step assembles the NEXT batch under its own backward pass
        (dmvae_plan_prefetch_batch: the gather rides on the idle CUs of the dZ launch, step_finalize one launch earlier) into the other
        of two batch buffers.  Two graphs, one per buffer, replayed in turn; the callable assembles the batch of the device cursor itself
        (one eager gather) whenever something other than its own replays has moved the cursor or touched the batch since (reset_epoch,
        an eager step, encode ...).  Bit-identical to the plain graph (tests/test_gpu_step.py).  Default: batches of at most
        PIPELINE_MAX_BATCH rows on plans that support it, no gradient exchange; DMVAE_PREFETCH=1 / 0 forces it on / off.  MEASURED
        (round 4, tools/knob_step.py <cfg>[:batch=N] pf 0 1 1 0, one engine, same buffers): 100 rows 0.1535 -> 0.1494 ms (-2.7 %; -3.4 % on
        another box); 256 / 512 / 1024 / 2048 rows -2.8 / -2.5 / -2.6 / -1.7 %; 4096 rows 0.2704 vs 0.2700 (nothing: measured with the 64-row dZ tiles, where the launch it saves, 5.9 us, comes back as +1.8 us in the dZ launch and a few
        tenths in most other kernels -- the batch is no longer fresh in the caches when the first layer reads it); 8192 rows, where the
        dZ launch has no idle CUs and the gather is a launch of its own mid-backward, 0.6085 vs 0.6109 (+0.4 %).  Since the thin dZ tiles
        (knob 18 = 2) the 4096-row dZ launch has no room for riders either: the gather is a launch of its own there too."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        saved = (self.param.clone(), self.m.clone(), self.v.clone(), self.state_t.clone(), getattr(self, "_moments_sharded", False))
        with torch.cuda.stream(side):
            for _ in range(2):   # warm-up outside capture (lazy module loads, RCCL channels)
                self.train_step(data, perm, None, None, None, 0, True, grad_sync, grad_scale, inv_B)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.param.copy_(saved[0]); self.m.copy_(saved[1]); self.v.copy_(saved[2]); self.state_t.copy_(saved[3])
        self.grad.zero_()
        # the whole fp32 arena has just been restored: whatever the sharded warm-up steps left stale outside this rank's slice is gone
        # (without this the guard of refresh_shadow refused -- every multi-rank bench.py / train_op run failed here: found by the
        # two-rank rehearsal, now tests/test_gpu_step.py::test_captured_step_under_the_sharded_exchange_two_ranks)
        self._master_stale = False
        self._moments_sharded = saved[4]             # (m, v are back to what they were before the warm-up steps)
        self.refresh_shadow()
        torch.cuda.synchronize(self.device)
        if pipelined is None:
            env = os.environ.get("DMVAE_PREFETCH", "")
            pipelined = env == "1" if env in ("0", "1") else self.max_batch <= PIPELINE_MAX_BATCH
        if grad_sync is None and pipelined and float(grad_scale) == 1.0 and self._prefetch_ok(data, perm, side):
            graphs = []
            for _ in range(2):                       # one graph per batch buffer; the plan is back on the first buffer afterwards
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self._prefetch_batch(data, perm)
                    self.forward_backward_update(None, None, None, inv_B)
                check(lib.dmvae_plan_swap_batch(self._plan), "dmvae_plan_swap_batch")
                graphs.append(g)
            self._graph = tuple(graphs)
            self._pf_primed, self._xsel = True, 0    # (_prefetch_ok left the batch of the device cursor in the first buffer; capturing ran nothing)

            def replay_pipelined():
                if not self._pf_primed:
                    self._load_batch_for_step(data, perm, 0, None, True)
                    self._pf_primed, self._xsel = True, 0
                graphs[self._xsel].replay()
                self._xsel ^= 1
            return replay_pipelined
        if grad_sync is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                self.train_step(data, perm, None, None, None, 0, True, None, grad_scale, inv_B)
            self._graph = (g,)

            def replay():                  # (a plain replay between pipelined replays of one engine moves the cursor under them)
                self._pf_primed = False
                g.replay()
            return replay
        # data parallel: issued eagerly.  MEASURED (one GPU, no-op exchange, `bench.py --dp-dry-run`): replaying the sequence as
        # several small graphs with the collectives between them was slower (three segments + three bucket updates: 0.362
        # vs 0.396 ms; one backward + Adam: 0.313 vs 0.326 ms) -- the host stays ahead of a 0.3 ms step and every graph
        # launch has its own cost; that form was removed.
        def eager_step():
            self.train_step(data, perm, None, None, None, 0, True, grad_sync, grad_scale, inv_B)
        self._graph = None
        return eager_step


def latent_eval(mean, log_var, logits, prior_means, prior_log_vars, eps=None, gumbel=None, mode="exact",
                temperature=1.0, kl_ratio=1.0, session=None):
    """Evaluate the latent-variable math of code/priors.py on the GPU through
    dmvae_latent_fwd: returns dict(Z, weights, kl_z, kl_c).  Arrays are
    numpy / array-like [B, D] / [B, K] / [K, D]."""
    sess = session or default_session()
    dev = sess.device
    f = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(dev)
    mean = np.asarray(mean, dtype=np.float32)
    logits = np.asarray(logits, dtype=np.float32)
    B, D = mean.shape
    K = logits.shape[1]
    Bp = (B + 63) // 64 * 64
    ldD, ldK = (D + 63) // 64 * 64, (K + 63) // 64 * 64

    def padded(a, ld):
        t = torch.zeros((Bp, ld), dtype=torch.float32, device=dev)
        a = f(a)
        t[: a.shape[0], : a.shape[1]] = a
        return t
    md, lvd, lgd = padded(mean, ldD), padded(log_var, ldD), padded(logits, ldK)
    epsd = f(eps) if eps is not None else torch.zeros((B, D), dtype=torch.float32, device=dev)
    gd = f(np.asarray(gumbel).reshape(B, K)) if gumbel is not None else torch.zeros((B, K), dtype=torch.float32, device=dev)
    pmd, plvd = f(prior_means), f(prior_log_vars)
    Z = torch.zeros((Bp, ldD), dtype=torch.float32, device=dev)
    w = torch.zeros((Bp, K), dtype=torch.float32, device=dev)
    gmu, glv, clv = (torch.zeros((Bp, ldD), dtype=torch.float32, device=dev) for _ in range(3))
    dlg = torch.zeros((Bp, ldK), dtype=torch.float32, device=dev)
    nblk = lib.dmvae_latent_nblocks_vade(Bp) if mode == "vade" else lib.dmvae_latent_nblocks(Bp, D, K)
    dpri = torch.zeros((nblk, 2 * K * D), dtype=torch.float32, device=dev)
    lp = torch.zeros((nblk, 2), dtype=torch.float32, device=dev)
    a = _lib.LatentArgs()
    a.B, a.B_pad, a.D, a.K = B, Bp, D, K
    a.mode, a.act_dtype = {"exact": 0, "relaxed": 1, "vade": 2}[mode], _lib.F32
    a.kl_ratio, a.temperature, a.inv_B = float(kl_ratio), float(temperature), 1.0 / B
    a.mean, a.ld_mean = md.data_ptr(), ldD
    a.log_var, a.ld_log_var = lvd.data_ptr(), ldD
    a.logits, a.ld_logits = lgd.data_ptr(), ldK
    a.eps, a.ld_eps = epsd.data_ptr(), D
    a.gumbel, a.ld_gumbel = gd.data_ptr(), K
    a.prior_means, a.prior_log_vars = pmd.data_ptr(), plvd.data_ptr()
    a.Z_act, a.ld_Z = Z.data_ptr(), ldD
    a.weights, a.ld_w = w.data_ptr(), K
    a.gmu, a.glv, a.clv, a.ld_g = gmu.data_ptr(), glv.data_ptr(), clv.data_ptr(), ldD
    a.dlogits_act, a.ld_dl = dlg.data_ptr(), ldK
    a.dprior_partials, a.loss_partials = dpri.data_ptr(), lp.data_ptr()
    check(lib.dmvae_latent_fwd(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.byref(a)), "dmvae_latent_fwd")
    torch.cuda.synchronize(dev)
    return dict(Z=Z[:B, :D].cpu().numpy(), weights=w[:B].cpu().numpy(),
                kl_z=float(lp[:, 0].double().sum().item() / B), kl_c=float(lp[:, 1].double().sum().item() / B))


def prof_enable(on=True):
    check(lib.dmvae_prof_enable(1 if on else 0), "dmvae_prof_enable")


def prof_collect(max_rows=64):
    rows = (_lib.ProfRow * max_rows)()
    n = lib.dmvae_prof_collect(rows, max_rows)
    out = []
    for i in range(n):
        r = rows[i]
        out.append(dict(name=r.name.decode(), launches=int(r.launches), total_ms=float(r.total_ms),
                        flops=float(r.flops), bytes=float(r.bytes), kernel_ms=float(r.kernel_ms),
                        kernel_launches=int(r.kernel_launches)))
    return out
