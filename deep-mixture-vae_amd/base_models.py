"""VAE / DeepMixtureVAE -- drop-in for the class surface of code/base_models.py
(:14-147 VAE, :150-432 DeepMixtureVAE) with the per-batch path executed by the
MI355X HIP library (dmvae_hip) instead of a TensorFlow session.

Same constructor arguments, method names, argument order and return types:
    DeepMixtureVAE(name, input_type, input_dim, latent_dim, n_classes,
                   activation=None, initializer=None, cnn=False).build_graph()
    .define_train_step(init_lr, decay_steps, decay_rate=0.9)
    .train_op(session, data, kl_ratio=1.0) -> float      (epoch-mean loss)
    .get_accuracy(session, data) -> float
    .sample_reparametrization_variables(n, variables=None) -> dict
    .sample_generative_feed(n, **kwargs) -> dict
`session` is a dmvae_hip.Session (device / stream / rank) or None.  What were
graph tensors to fetch (mean, log_var, logits, cluster_probs, Z, decoded_X,
reconstructed_X) are methods evaluating on demand: encode(X), decode(Z),
reconstruct(X, epsilon=None).  Extensions are keyword-only and default to the
reference's behaviour: batch_size (reference hard-codes 100, train.py:215-216),
dtype ("bf16" throughput / "fp32" parity), enc_layers / head_dim / dec_layers
(reference literals 500,500 / 2000 / 2000,500,500), gumbel + temperature
(Gumbel-Softmax relaxed KL; default off = the live graph),
noise ("device" Philox | "host" NumPy stream of the reference), seed.
"""
import math
import os
import zipfile

import numpy as np

import priors
from includes.network import DeepNetwork
from includes.utils import get_clustering_accuracy


class VAE:
    def __init__(self, name, input_type, input_dim, latent_dim, activation=None, initializer=None):
        self.name = name
        self.input_dim = input_dim
        self.latent_dim = latent_dim
        self.input_type = input_type
        self.activation = activation
        self.initializer = initializer
        self.path = ""
        self.kl_ratio = 1.0          # placeholder_with_default(1.0), base_models.py:28-30
        self.is_training = True      # placeholder_with_default(True), :32-34 (no BN on the path: unused)
        self.X = None
        self.decoded_X = None
        self.train_step = None
        self.latent_variables = dict()

    def build_graph(self, encoder_layer_sizes, decoder_layer_sizes):
        raise NotImplementedError

    def sample_reparametrization_variables(self, n, variables=None):
        """base_models.py:44-56: {epsilon placeholder name: host noise}; draws in
        the dict order of latent_variables (C then Z) from the global NumPy RNG."""
        samples = dict()
        if variables is None:
            for lv, eps, _ in self.latent_variables.values():
                if eps is not None:
                    samples[eps] = lv.sample_reparametrization_variable(n)
        else:
            for var in variables:
                lv, eps, _ = self.latent_variables[var]
                if eps is not None:
                    samples[eps] = lv.sample_reparametrization_variable(n)
        return samples

    def sample_generative_feed(self, n, **kwargs):
        samples = dict()
        for name, (lv, _, _) in self.latent_variables.items():
            kwargs_ = dict() if name not in kwargs else kwargs[name]
            samples[name] = lv.sample_generative_feed(n, **kwargs_)
        return samples

    # the three loss definitions exist as graph-building steps in the reference
    # (:66-93); here the loss is computed inside the fused step, nothing to build
    def define_latent_loss(self):
        self.latent_loss = "KL_C + KL_Z (dmvae_latent_fwd)"

    def define_recon_loss(self):
        if self.input_type not in ("binary", "real"):
            raise NotImplementedError
        self.recon_loss = "reconstruction loss (DMVAE_EPI_BIAS_RECON)"

    def define_train_loss(self):
        self.define_latent_loss()
        self.define_recon_loss()
        self.loss = "recon + kl_ratio * latent"

    def define_train_step(self, init_lr, decay_steps, decay_rate=0.9):
        """base_models.py:95-110.  exponential_decay is called with a literal
        global_step=0, so the learning rate is the constant init_lr;
        decay_steps / decay_rate are accepted and, as in the reference, inert."""
        self.define_train_loss()
        self._lr = float(init_lr)
        self._engine.write_state(lr=self._lr)
        self.train_step = "adam_tf"

    def debug(self, session, data):
        import pdb
        for batch in data.get_batches():
            feed = {"X": batch}
            feed.update(self.sample_reparametrization_variables(len(batch)))
            pdb.set_trace()
            break


def _cnn_encoder_network(out_dim):
    """the convolutional encoder as the reference specifies it (base_models.py:179-205 with ("fc", 2048 -> 500) for DMVAE,
    :459-488 with ("fc", 2048 -> 128) for VaDE): the DeepNetwork spec list drives the plan's CNN trunk -- the plan implements
    exactly this stack, any other list is refused here rather than silently ignored"""
    from dmvae_hip.runtime import CONV_STACK, CONV_FLAT
    enc_spec = [("cn", {"n_kernels": 32, "prev_n_kernels": 1, "kernel": (3, 3)}),
                ("cn", {"n_kernels": 32, "prev_n_kernels": 32, "kernel": (3, 3)}), ("mp", {"k": 2}),
                ("cn", {"n_kernels": 64, "prev_n_kernels": 32, "kernel": (3, 3)}),
                ("cn", {"n_kernels": 64, "prev_n_kernels": 64, "kernel": (3, 3)}), ("mp", {"k": 2}),
                ("cn", {"n_kernels": 128, "prev_n_kernels": 64, "kernel": (3, 3)}),
                ("cn", {"n_kernels": 128, "prev_n_kernels": 128, "kernel": (3, 3)}), ("mp", {"k": 2}),
                ("fc", {"input_dim": 2048, "output_dim": out_dim})]
    net = DeepNetwork("layers", enc_spec, activation="relu", initializer="xavier")
    stack, side = net.conv_stack(28)
    if stack != tuple((ci, co, hw, pool) for _, ci, co, hw, pool in CONV_STACK) or side * side * stack[-1][1] != CONV_FLAT:
        raise NotImplementedError("the plan's CNN trunk is the stack of base_models.py:181-202; got %r" % (stack,))
    assert net.widths() == (out_dim,)
    return net


class DeepMixtureVAE(VAE):
    def __init__(self, name, input_type, input_dim, latent_dim, n_classes, activation=None, initializer=None,
                 cnn=False, *, batch_size=100, dtype="bf16", enc_layers=(500, 500), head_dim=2000,
                 dec_layers=(2000, 500, 500), gumbel=False, temperature=1.0, noise="device", seed=0,
                 deterministic=True, session=None):
        VAE.__init__(self, name, input_type, input_dim, latent_dim, activation=activation, initializer=initializer)
        self.n_classes = n_classes
        # The checked-in reference forces cnn = True (base_models.py:156); the path BASELINE.json names is
        # the MLP branch (:218-226), the default here.  cnn=True builds the checked-in trunk
        # (:176-216): six 3x3 SAME convolutions, three 2x2 SAME max-pools, FullyConnected 2048 -> 500.
        self.cnn = bool(cnn)
        if activation not in (None, "relu") and getattr(activation, "__name__", "") != "relu":
            raise NotImplementedError("activation must be ReLU (train.py:196 passes tf.nn.relu)")
        if noise not in ("device", "host"):
            raise ValueError("noise must be 'device' or 'host'")
        self.batch_size = int(batch_size)
        self.dtype = dtype
        self.enc_layers, self.head_dim, self.dec_layers = tuple(enc_layers), int(head_dim), tuple(dec_layers)
        if self.cnn:
            if int(input_dim) != 784:
                raise ValueError("cnn=True reshapes the inputs to 28x28x1 (base_models.py:176): input_dim must be 784")
            self.enc_layers = (self.enc_layers[-1],)      # the trunk's one dense layer: ("fc", 2048 -> 500), :202
        self.gumbel, self.temperature = bool(gumbel), float(temperature)
        self.noise, self.seed, self.deterministic = noise, int(seed), bool(deterministic)
        self._session = session
        self._engine = None
        self._replay = None
        self._replay_key = None
        self._perm = None

    # ------------------------------------------------------------------ graph
    def build_graph(self):
        from dmvae_hip import StepEngine, default_session
        sess = self._session or default_session()
        self._session = sess
        # decoder spec exactly as the reference writes it (base_models.py:280-288)
        prev, spec = self.latent_dim, []
        for w in self.dec_layers:
            spec.append(("fc", {"input_dim": prev, "output_dim": w}))
            prev = w
        self.decoder_network = DeepNetwork("layers", spec, activation="relu", initializer="xavier")
        assert self.decoder_network.widths() == self.dec_layers
        if self.cnn:
            self.encoder_network = _cnn_encoder_network(self.enc_layers[0])
        self._engine = StepEngine(self.input_dim, self.latent_dim, self.n_classes, enc_layers=self.enc_layers,
                                  head_dim=self.head_dim, dec_layers=self.dec_layers, input_type=self.input_type,
                                  dtype=self.dtype, max_batch=self.batch_size, mode="relaxed" if self.gumbel else "exact",
                                  temperature=self.temperature, seed=self.seed + 7919 * sess.rank,
                                  deterministic=self.deterministic, session=sess, cnn=self.cnn)
        self._engine.init_parameters(self.seed)
        # names of the graph's placeholders / tensors (fetch through the methods below)
        self.X, self.epsilon, self.cluster = "X", "epsilon_Z", "epsilon_C"
        self.mean, self.log_var, self.logits = "mean", "log_var", "logits"
        self.cluster_probs, self.Z = "cluster_probs", "Z"
        self.decoded_X, self.reconstructed_X = "decoded_X", "reconstructed_X"
        self.latent_variables = dict()
        self.latent_variables.update({
            "C": (priors.DiscreteFactorial("cluster", 1, self.n_classes), self.cluster, {"logits": self.logits}),
            "Z": (priors.NormalMixtureFactorial("representation", self.latent_dim, self.n_classes, engine=self._engine),
                  self.epsilon,
                  {"mean": self.mean, "log_var": self.log_var, "weights": self.cluster_probs,
                   "cluster_sample": self.gumbel}),
        })
        return self

    @property
    def engine(self):
        return self._engine

    # ------------------------------------------------------------------ training
    def _epoch_perm(self, data, sess):
        """the epoch's row order -> this rank's slice of every global batch, on the device.
        Returns (perm, n_full, tail, epoch_weight) -- see dmvae_hip.parallel.epoch_plan: with more than
        one rank only the full global batches are trained, the step count is the same on every rank."""
        import torch
        from dmvae_hip.parallel import epoch_plan
        order, n_full, tail, weight = epoch_plan(data.reshuffle(), data.batch_size, sess.rank, sess.world_size)
        t = torch.as_tensor(np.ascontiguousarray(order, dtype=np.int32))
        if self._perm is None or self._perm.numel() != t.numel():
            self._perm = torch.empty(t.numel(), dtype=torch.int32, device=sess.device)
        self._perm.copy_(t)
        return self._perm, n_full, tail, weight

    def train_op(self, session, data, kl_ratio=1.0):
        """One epoch, base_models.py:112-132: for every batch of data.get_batches()
        run [loss, train_step]; return sum(batch_loss) / epoch_len.  The loss is
        accumulated on the device and read back once per epoch."""
        assert(self.train_step is not None)
        import torch
        from dmvae_hip import make_exchange
        sess = session or self._session
        eng = self._engine
        world = sess.world_size
        gb = data.batch_size                                  # global batch
        if gb % world:
            raise ValueError("batch_size %d is not divisible by the world size %d" % (gb, world))
        b = gb // world                                       # per-rank batch
        if b != eng.max_batch:
            raise ValueError("per-rank batch %d != the size the model was built for (%d)" % (b, eng.max_batch))
        rows = data.device_rows(sess.device)
        perm, n_full, tail, weight = self._epoch_perm(data, sess)
        import time
        torch.cuda.synchronize(sess.device)
        self._epoch_t0 = time.perf_counter()                   # (behind the row upload and the epoch's permutation: the steps themselves)
        ex = make_exchange(4 * eng.param.numel())
        sync = ex if ex.enabled else None
        eng.reset_epoch(n_full + (1 if tail else 0), kl_ratio=kl_ratio, epoch_weight=weight)
        host_noise = self.noise == "host"
        if host_noise and world > 1:
            raise NotImplementedError("noise='host' replays the reference's single-process NumPy stream; use noise='device' with more than one rank")

        def host_feed(n):
            feed = self.sample_reparametrization_variables(n)        # C (gumbel) first, then Z: reference order
            eps = torch.as_tensor(np.ascontiguousarray(feed[self.epsilon], dtype=np.float32)).to(sess.device)
            g = None
            if self.gumbel:
                g = torch.as_tensor(np.ascontiguousarray(feed[self.cluster].reshape(n, self.n_classes), dtype=np.float32)).to(sess.device)
            return eps, g

        if host_noise:
            for i in range(n_full):
                eps, g = host_feed(b)
                eng.train_step(rows, perm, b, eps, g, first=i * b, grad_sync=sync, grad_scale=ex.grad_scale)
        else:
            # the captured graph holds the device pointers of the rows and of the row order: recapture
            # when either buffer (another Dataset, a re-upload) or the exchange changes
            key = (rows.data_ptr(), perm.data_ptr(), b, sync is None)
            if n_full > 0 and (self._replay is None or self._replay_key != key):
                self._replay = eng.capture_step(rows, perm, grad_sync=sync, grad_scale=ex.grad_scale)
                self._replay_key = key
                eng.reset_epoch(n_full + (1 if tail else 0), kl_ratio=kl_ratio, epoch_weight=weight)
            for _ in range(n_full):
                self._replay()
        if tail:                                             # the short last batch (utils.py:462-463), issued eagerly
            eps, g = host_feed(tail) if host_noise else (None, None)
            inv_B = 1.0 / tail
            eng.train_step(rows, perm, tail, eps, g, first=n_full * b, grad_sync=sync, grad_scale=ex.grad_scale, inv_B=inv_B)
        import time
        t_wall = getattr(self, "_epoch_t0", None)
        torch.cuda.synchronize(sess.device)
        st = eng.read_state()
        loss = float(st.epoch_loss)
        terms = [loss, float(st.epoch_recon), float(st.epoch_klz), float(st.epoch_klc)]
        if world > 1:
            terms = ex.mean_scalars(terms)
            loss = terms[0]
        # what the epoch was made of (the reference prints only the loss; train.py writes these to its metrics log)
        n_rows = n_full * gb + tail
        self.last_epoch = dict(loss=terms[0], recon=terms[1], kl_z=terms[2], kl_c=terms[3], kl_ratio=float(kl_ratio), rows=int(n_rows),
                               steps=int(n_full + (1 if tail else 0)), seconds=(time.perf_counter() - t_wall) if t_wall is not None else None)
        return loss

    # ------------------------------------------------------------------ inference pieces
    def _batches(self, X):
        import torch
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float32))
        dev = self._session.device
        b = self._engine.max_batch
        for s in range(0, len(X), b):
            yield s, torch.as_tensor(X[s:s + b]).to(dev)

    def encode(self, X):
        """(mean, log_var, logits) of q(z|x), q(c|x): base_models.py:229-248."""
        eng = self._engine
        out = [[], [], []]
        for _, xb in self._batches(X):
            n = xb.shape[0]
            eng.load_batch(xb, None, 0, n)
            eng.encode(n)
            for k, name in enumerate(("mean", "log_var", "logits")):
                out[k].append(eng.view(name, n).cpu().numpy().copy())
        return tuple(np.concatenate(o, axis=0) for o in out)

    def decode(self, Z):
        """reconstructed_X for given Z (what visualization.py:83-87 fetches by feeding model.Z)."""
        import torch
        eng = self._engine
        Z = np.ascontiguousarray(np.asarray(Z, dtype=np.float32))
        out = []
        for s in range(0, len(Z), eng.max_batch):
            zb = torch.as_tensor(Z[s:s + eng.max_batch]).to(self._session.device)
            eng.decode(zb)
            out.append(eng.view("recon", zb.shape[0]).cpu().numpy().copy())
        return np.concatenate(out, axis=0)

    def reconstruct(self, X, epsilon=None):
        """reconstructed_X with the given noise (zeros by default, as
        visualization.py:39-46 feeds): Z = mean + exp(log_var/2) * epsilon."""
        mean, log_var, _ = self.encode(X)
        Z = mean if epsilon is None else mean + np.exp(log_var / 2) * np.asarray(epsilon)
        return self.decode(Z)

    def get_accuracy(self, session, data):
        """base_models.py:425-432: logits of every batch -> clustering accuracy."""
        order = data.reshuffle()               # the reference iterates data.get_batches(), which reshuffles
        _, _, logits = self.encode(data._rows[order])
        return get_clustering_accuracy(logits, data._cls[order])

    # ------------------------------------------------------------------ pretraining (base_models.py:304-423)
    # the variables tf.get_collection(TRAINABLE_VARIABLES, scope=name + "/encoder_network/c") returns (:313-315)
    PRIOR_VAR_LIST = ("W_ch", "b_ch", "W_logits", "b_logits")

    def define_pretrain_step(self, vae_lr, prior_lr):
        """base_models.py:304-320: two more Adam optimizers, each with its own slots --
        vae_train_step = Adam(vae_lr).minimize(recon_loss) over every variable the
        reconstruction depends on, prior_train_step = Adam(prior_lr).minimize(latent_loss,
        var_list = the c-head)."""
        self.define_train_loss()
        self.vae_loss = "recon"
        self._vae_lr, self._prior_lr = float(vae_lr), float(prior_lr)
        self.vae_train_step = "adam_tf(recon_loss)"
        self.prior_train_step = "adam_tf(latent_loss, var_list=encoder_network/c)"

    def _ckpt(self, stage):
        return os.path.join(self.path, stage, "parameters.npz") if self.path else None

    def _restore(self, stage):
        """the reference wraps its restore in try / except and carries on (base_models.py:324-329, :354-359)"""
        path = self._ckpt(stage)
        if not (path and os.path.exists(path)):
            return False
        try:
            with np.load(path, allow_pickle=False) as f:
                self.load_state_dict({k: f[k] for k in f.files})
            return True
        except (OSError, ValueError, KeyError, RuntimeError, EOFError, zipfile.BadZipFile) as e:
            print("Could not read %s: %s" % (path, e))
            return False

    def _save(self, stage):
        """rank 0 writes (to a temporary file, then an atomic rename); the other ranks wait at a barrier
        so that none of them reads a half-written archive"""
        path = self._ckpt(stage)
        if not path:
            return
        sess = self._session
        if sess is not None and sess.world_size > 1:      # sharded bf16 steps leave a rank's fp32 weights current on its own slice only
            from dmvae_hip import make_exchange
            self._engine.sync_master(make_exchange(4 * self._engine.param.numel()))      # collective: every rank is here
        if sess is None or sess.rank == 0:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = path + ".tmp.npz"
            np.savez(tmp, **self.state_dict())
            os.replace(tmp, path)
        if sess is not None and sess.world_size > 1:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.barrier()

    def _pretrain_epoch(self, sess, data, stage):
        """one epoch of a pretraining stage: epsilon = 0 (so Z = mean), the stage's loss and
        variable list; returns sum(batch_loss) / epoch_len like the reference's loops (:331-350, :395-414)."""
        import torch
        from dmvae_hip import make_exchange
        eng = self._engine
        world = sess.world_size
        b = data.batch_size // world
        if data.batch_size % world or b != eng.max_batch:
            raise ValueError("batch_size %d does not match the model (%d per rank x %d ranks)" % (data.batch_size, eng.max_batch, world))
        rows = data.device_rows(sess.device)
        perm, n_full, tail, weight = self._epoch_perm(data, sess)
        import time
        torch.cuda.synchronize(sess.device)
        self._epoch_t0 = time.perf_counter()                   # (behind the row upload and the epoch's permutation: the steps themselves)
        ex = make_exchange(4 * eng.param.numel())
        sync = ex if ex.enabled else None
        # recon-only objective = the full loss at kl_ratio 0: every KL gradient carries the factor r
        eng.reset_epoch(n_full + (1 if tail else 0), kl_ratio=0.0 if stage == "vae" else 1.0, epoch_weight=weight)
        zeros = {}

        def noise(n):
            if n not in zeros:
                zeros[n] = (torch.zeros((n, self.latent_dim), dtype=torch.float32, device=sess.device),
                            torch.zeros((n, self.n_classes), dtype=torch.float32, device=sess.device) if self.gumbel else None)
            return zeros[n]
        frozen = [k for k in eng.parameter_names() if k not in self.PRIOR_VAR_LIST]
        for i in range(n_full + (1 if tail else 0)):
            n = b if i < n_full else tail
            inv_B = (world / float(data.batch_size)) if i < n_full else 1.0 / n
            eps, g = noise(n)
            if stage == "vae":
                eng.train_step(rows, perm, n, eps, g, first=i * b, grad_sync=sync, grad_scale=ex.grad_scale, inv_B=inv_B)
            else:
                eng.load_batch(rows, perm, i * b, n)
                eng.forward_backward(n, eps, g, inv_B)
                if sync is not None:
                    sync(eng.grad)
                for k in frozen:                      # minimize(..., var_list=c-head): nothing else moves
                    eng.grad_view(k).zero_()
                eng.update(ex.grad_scale)
        torch.cuda.synchronize(sess.device)
        st = eng.read_state()
        loss = float(st.epoch_recon) if stage == "vae" else float(st.epoch_klz) + float(st.epoch_klc)
        if world > 1:
            loss = ex.mean_scalars([loss])[0]
        self._replay = None        # the step graph was captured for the main optimizer's state; recapture after pretraining
        return loss

    def _sync_master(self, sess):
        """COLLECTIVE under data parallelism (every rank calls it at the same point): bring the fp32 master weights up to date on
        every rank after sharded bf16 steps, before anything reads or rewrites the whole fp32 arena (StepEngine.sync_master)."""
        if sess is not None and sess.world_size > 1:
            from dmvae_hip import make_exchange
            self._engine.sync_master(make_exchange(4 * self._engine.param.numel()))

    def pretrain_vae(self, session, data, n_epochs):
        """base_models.py:322-350."""
        sess = session or self._session
        if not self._restore("vae"):
            print("Could not load trained ae parameters")
        self._engine.reset_optimizer(self._vae_lr)
        min_loss = float("inf")
        for _ in range(n_epochs):
            loss = self._pretrain_epoch(sess, data, "vae")
            if loss <= min_loss:
                min_loss = loss
                self._save("vae")
        self._sync_master(sess)       # the next stage sets parameters / runs the replicated Adam over the whole fp32 arena
        return min_loss

    def pretrain_prior(self, session, data, n_epochs):
        """base_models.py:352-414: prior tables from a diagonal GMM on the encoder means, then
        Adam on the latent loss over the c-head only."""
        sess = session or self._session
        if not self._restore("prior"):
            print("Could not load trained prior parameters")
            if n_epochs > 0:
                from sklearn.mixture import GaussianMixture
                Z = self.encode(data.data)[0]
                gmm_model = GaussianMixture(n_components=self.n_classes, covariance_type="diag", max_iter=n_epochs,
                                            n_init=20, weights_init=np.ones(self.n_classes) / self.n_classes)
                gmm_model.fit(Z)
                self._engine.set_parameters({"prior_means": gmm_model.means_,
                                             "prior_log_vars": np.log(gmm_model.covariances_ + 1e-20)})
                self._save("prior")
        self._engine.reset_optimizer(self._prior_lr)
        min_loss = float("inf")
        for _ in range(n_epochs):
            loss = self._pretrain_epoch(sess, data, "prior")
            if loss <= min_loss:
                min_loss = loss
                self._save("prior")
        return min_loss

    def pretrain(self, session, data, n_epochs_vae, n_epochs_gmm):
        """base_models.py:416-423."""
        assert(self.vae_train_step is not None and self.prior_train_step is not None)
        self.pretrain_vae(session, data, n_epochs_vae)
        self.pretrain_prior(session, data, n_epochs_gmm)
        # the main optimizer (define_train_step) is its own AdamOptimizer: fresh slots, its own learning rate
        self._engine.reset_optimizer(getattr(self, "_lr", None))

    # ------------------------------------------------------------------ checkpoint (trainables only, like tf.train.Saver)
    def state_dict(self):
        return {k: v for k, v in self._engine.get_parameters().items()}

    def load_state_dict(self, sd):
        self._engine.set_parameters(sd)


class VaDE(DeepMixtureVAE):
    """code/base_models.py:435-670 on the same HIP kernels: one encoder of FullyConnected layers
    784 -> 2000 -> 500 -> 500 (:490-499), mean / log_var linear straight off it (:501-507), q(c|x) := p(c|z) =
    get_cluster_probs(Z) (:526-527, priors.py:91-102) as the mixture weights of the exact KL and as the probabilities of
    the categorical KL, decoder D -> 500 -> 500 -> 2000 -> input (:530-547).  The per-batch step is the DMVAE step plan
    with dmvae_config.model = DMVAE_MODEL_VADE: no head hidden layers / logits, latent stage = dmvae_latent_fwd mode 2
    (csrc/latent_vade.hip), whose gradients include the path through the responsibilities into Z and the prior tables.
    cnn=True (:456-488): the batch as 28x28x1 images through the same six-convolution / three-pool stack as DMVAE's
    checked-in encoder, ending in ("fc", 2048 -> 128), with mean / log_var straight off those 128 units
    (dmvae_config.trunk = DMVAE_TRUNK_CNN, enc = (128,))."""

    def __init__(self, name, input_type, input_dim, latent_dim, n_classes, activation=None, initializer=None, cnn=False, *,
                 batch_size=100, dtype="bf16", enc_layers=(2000, 500, 500), dec_layers=(500, 500, 2000), noise="device", seed=0,
                 deterministic=True, session=None):
        if cnn and tuple(enc_layers) == (2000, 500, 500):
            enc_layers = (128,)            # base_models.py:486: ("fc", {"input_dim": 2048, "output_dim": 128})
        DeepMixtureVAE.__init__(self, name, input_type, input_dim, latent_dim, n_classes, activation=activation, initializer=initializer,
                                cnn=cnn, batch_size=batch_size, dtype=dtype, enc_layers=enc_layers, head_dim=64, dec_layers=dec_layers,
                                gumbel=False, temperature=1.0, noise=noise, seed=seed, deterministic=deterministic, session=session)

    def build_graph(self):
        from dmvae_hip import StepEngine, default_session
        sess = self._session or default_session()
        self._session = sess
        enc_spec, prev = [], self.input_dim                   # the two DeepNetwork spec lists exactly as the reference writes them
        for w in self.enc_layers:
            enc_spec.append(("fc", {"input_dim": prev, "output_dim": w}))
            prev = w
        dec_spec, prev = [], self.latent_dim
        for w in self.dec_layers:
            dec_spec.append(("fc", {"input_dim": prev, "output_dim": w}))
            prev = w
        self.encoder_network = (_cnn_encoder_network(self.enc_layers[0]) if self.cnn else
                                DeepNetwork("layers", enc_spec, activation="relu", initializer="xavier"))
        self.decoder_network = DeepNetwork("layers", dec_spec, activation="relu", initializer="xavier")
        self._engine = StepEngine(self.input_dim, self.latent_dim, self.n_classes, enc_layers=self.enc_layers, head_dim=64,
                                  dec_layers=self.dec_layers, input_type=self.input_type, dtype=self.dtype, max_batch=self.batch_size,
                                  mode="exact", seed=self.seed + 7919 * sess.rank, deterministic=self.deterministic, session=sess,
                                  model="vade", cnn=self.cnn)
        self._engine.init_parameters(self.seed)
        self.X, self.epsilon = "X", "epsilon"
        self.mean, self.log_var, self.cluster_probs, self.Z = "mean", "log_var", "cluster_probs", "Z"
        self.decoded_X, self.reconstructed_X = "decoded_X", "reconstructed_X"
        self.latent_variables = dict()
        self.latent_variables.update({
            "Z": (priors.NormalMixtureFactorial("representation", self.latent_dim, self.n_classes, engine=self._engine), self.epsilon,
                  {"mean": self.mean, "log_var": self.log_var, "cluster_sample": False, "weights": self.cluster_probs}),
        })
        self.latent_variables.update({
            "C": (priors.DiscreteFactorial("cluster", 1, self.n_classes), None, {"probs": self.cluster_probs}),
        })
        return self

    def encode(self, X):
        """(mean, log_var) of q(z|x), base_models.py:501-507 (VaDE has no logits: q(c|x) comes from the sample)."""
        mean, log_var, _ = DeepMixtureVAE.encode(self, X)
        return mean, log_var

    def reconstruct(self, X, epsilon=None):
        mean, log_var = self.encode(X)
        Z = mean if epsilon is None else mean + np.exp(log_var / 2) * np.asarray(epsilon)
        return self.decode(Z)

    def cluster_probabilities(self, X, epsilon):
        """fetching `cluster_probs` with X and epsilon fed (base_models.py:659-663)"""
        mean, log_var = self.encode(X)
        Z = mean + np.exp(log_var / 2) * np.asarray(epsilon, dtype=np.float32)
        return self.latent_variables["Z"][0].get_cluster_probs(Z)

    # ------------------------------------------------------------------ pretraining (base_models.py:573-652)
    def define_pretrain_step(self, vae_lr, _prior_lr=None):
        self.define_train_loss()
        self.vae_loss = "recon"
        self._vae_lr, self._prior_lr = float(vae_lr), None
        self.vae_train_step = "adam_tf(recon_loss)"
        self.prior_train_step = None

    def pretrain_prior(self, session, data, n_epochs):
        """base_models.py:611-646: only the GMM initialisation of the prior tables (n_init = 5), no Adam stage."""
        if not self._restore("prior"):
            print("Could not load pretrained prior parameters")
            if n_epochs > 0:
                from sklearn.mixture import GaussianMixture
                Z = self.encode(data.data)[0]
                gmm_model = GaussianMixture(n_components=self.n_classes, covariance_type="diag", max_iter=n_epochs,
                                            n_init=5, weights_init=np.ones(self.n_classes) / self.n_classes)
                gmm_model.fit(Z)
                self._engine.set_parameters({"prior_means": gmm_model.means_,
                                             "prior_log_vars": np.log(gmm_model.covariances_ + 1e-20)})
                self._save("prior")

    def pretrain(self, session, data, n_epochs_vae, n_epochs_prior):
        assert(self.vae_train_step is not None)
        self.pretrain_vae(session, data, n_epochs_vae)
        self.pretrain_prior(session, data, n_epochs_prior)
        self._engine.reset_optimizer(getattr(self, "_lr", None))

    def get_accuracy(self, session, data, k=10):
        """base_models.py:654-670: cluster_probs averaged over k noise draws (the reference's NumPy stream, one
        sample_reparametrization_variables(n, ["Z"]) per draw over the whole set), then the clustering accuracy."""
        X = data.data
        weights = []
        for _ in range(k):
            feed = self.sample_reparametrization_variables(len(X), variables=["Z"])
            weights.append(self.cluster_probabilities(X, feed[self.epsilon]))
        weights = np.mean(np.array(weights), axis=0)
        return get_clustering_accuracy(weights, data.classes)
