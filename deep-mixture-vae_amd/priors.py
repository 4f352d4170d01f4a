"""Latent-variable classes of the DMVAE drop-in -- same names, constructor
arguments and method protocol as code/priors.py of the reference.

In the reference every method builds TensorFlow graph nodes.  Here the
arithmetic of kl_from_prior / inverse_reparametrize is evaluated by the fused
HIP latent kernel (dmvae_latent_fwd) on array inputs and returned as NumPy;
inside the training step the same kernel runs fused with its gradients and is
never called through these classes.  Host noise samplers draw from the global
NumPy RNG exactly as the reference does.
"""
import numpy as np

from includes.utils import sample_gumbel


class LatentVariable:
    def kl_from_prior(self, **kwargs):
        raise NotImplementedError

    def sample_reparametrization_variable(self, **kwargs):
        raise NotImplementedError

    def sample_generative_feed(self, **kwargs):
        raise NotImplementedError

    def inverse_reparametrize(self, **kwargs):
        raise NotImplementedError


def _latent_eval(*a, **k):
    from dmvae_hip import latent_eval   # imported late: needs the GPU library
    return latent_eval(*a, **k)


class NormalFactorial(LatentVariable):
    """code/priors.py:21-47 -- the K = 1, N(0, I) special case."""

    def __init__(self, name, dim):
        self.name = name
        self.dim = dim

    def sample_reparametrization_variable(self, n):
        return np.random.randn(n, self.dim)

    def sample_generative_feed(self, n, **kwargs):
        return np.random.randn(n, self.dim)

    def inverse_reparametrize(self, epsilon, parameters):
        assert("mean" in parameters and "log_var" in parameters)
        z = np.zeros((1, self.dim), np.float32)
        return _latent_eval(parameters["mean"], parameters["log_var"], np.zeros((len(epsilon), 1), np.float32),
                            z, z, eps=epsilon)["Z"]

    def kl_from_prior(self, parameters, eps=1e-20):
        assert("mean" in parameters and "log_var" in parameters)
        z = np.zeros((1, self.dim), np.float32)
        mean = np.asarray(parameters["mean"])
        return _latent_eval(mean, parameters["log_var"], np.zeros((len(mean), 1), np.float32), z, z)["kl_z"]


class NormalMixtureFactorial(LatentVariable):
    """code/priors.py:50-147.  `means` / `log_vars` are the trainable prior
    tables; when the object belongs to a built model they are live views of
    the parameter arena on the GPU, otherwise NumPy arrays initialised like
    the reference (random_normal / zeros, priors.py:57-65)."""

    def __init__(self, name, dim, n_classes, trainable=True, engine=None):
        self.name = name
        self.dim = dim
        self.n_classes = n_classes
        self.trainable = trainable
        self._engine = engine
        if engine is None:
            self._means = np.random.randn(n_classes, dim).astype(np.float32)
            self._log_vars = np.zeros((n_classes, dim), np.float32)

    @property
    def means(self):
        if self._engine is not None:
            return self._engine.param_view("prior_means").detach().cpu().numpy()
        return self._means

    @means.setter
    def means(self, v):
        if self._engine is not None:
            self._engine.set_parameters({"prior_means": v})
        else:
            self._means = np.asarray(v, np.float32)

    @property
    def log_vars(self):
        if self._engine is not None:
            return self._engine.param_view("prior_log_vars").detach().cpu().numpy()
        return self._log_vars

    @log_vars.setter
    def log_vars(self, v):
        if self._engine is not None:
            self._engine.set_parameters({"prior_log_vars": v})
        else:
            self._log_vars = np.asarray(v, np.float32)

    def sample_reparametrization_variable(self, n):
        return np.random.randn(n, self.dim)

    def sample_generative_feed(self, n, **kwargs):
        """z ~ N(mu_c, sigma_c^2) (priors.py:70-84); `session` is accepted for
        signature compatibility and not needed."""
        samples = np.random.randn(n, self.dim)
        if "c" not in kwargs:
            c = np.random.randint(0, 10, n, dtype=np.int32)
        else:
            c = kwargs["c"]
        means, log_vars = self.means[c, :], self.log_vars[c, :]
        return means + samples * np.exp(log_vars / 2.0)

    def inverse_reparametrize(self, epsilon, parameters):
        assert("mean" in parameters and "log_var" in parameters)
        mean = np.asarray(parameters["mean"])
        return _latent_eval(mean, parameters["log_var"], np.zeros((len(mean), self.n_classes), np.float32),
                            self.means, self.log_vars, eps=epsilon)["Z"]

    def get_cluster_probs(self, Z):
        """priors.py:91-102: softmax_k of -1/2 [sum_d (z - mu_k)^2 / sigma_k^2 + sum_d log sigma_k^2] -- what VaDE uses as
        q(c|x) (base_models.py:526-527).  Evaluated by the VaDE mode of the HIP latent kernel (dmvae_latent_fwd mode 2,
        csrc/latent_vade.hip) with mean = Z and epsilon = 0, so that its sample IS Z."""
        Z = np.asarray(Z, dtype=np.float32)
        zeros = np.zeros_like(Z)
        return _latent_eval(Z, zeros, np.zeros((len(Z), self.n_classes), np.float32), self.means, self.log_vars,
                            eps=zeros, mode="vade")["weights"]

    def kl_from_prior(self, parameters, eps=1e-20):
        assert(
            "cluster_sample" in parameters and
            "weights" in parameters and
            "log_var" in parameters and
            "mean" in parameters
        )
        w = np.reshape(np.asarray(parameters["weights"], dtype=np.float64), (-1, self.n_classes))
        # the kernel takes logits: softmax(log w) == w for a normalised w
        logits = np.log(np.maximum(w, 1e-38))
        mode = "relaxed" if parameters["cluster_sample"] else "exact"
        return _latent_eval(parameters["mean"], parameters["log_var"], logits, self.means, self.log_vars,
                            mode=mode, temperature=1.0)["kl_z"]


class DiscreteFactorial(LatentVariable):
    """code/priors.py:150-201 with dim = 1 (the DMVAE cluster variable)."""

    def __init__(self, name, dim, n_classes):
        if dim != 1:
            raise NotImplementedError("DiscreteFactorial: the DMVAE path uses dim = 1")
        self.name = name
        self.dim = dim
        self.n_classes = n_classes

    def sample_reparametrization_variable(self, n):
        return sample_gumbel((n, self.dim, self.n_classes))

    def sample_generative_feed(self, n, **kwargs):
        samples = sample_gumbel((n, self.dim, self.n_classes))
        samples = np.reshape(samples, (-1, self.n_classes))
        samples = np.asarray(np.equal(samples, np.max(samples, 1, keepdims=True)), dtype=samples.dtype)
        return np.reshape(samples, (-1, self.dim, self.n_classes))

    def inverse_reparametrize(self, epsilon, parameters):
        assert("logits" in parameters and "temperature" in parameters)
        logits = np.reshape(np.asarray(parameters["logits"]), (-1, self.n_classes))
        g = np.reshape(np.asarray(epsilon), (-1, self.n_classes))
        z = np.zeros((self.n_classes, 1), np.float32)
        res = _latent_eval(np.zeros((len(logits), 1), np.float32), np.zeros((len(logits), 1), np.float32), logits,
                           z, z, gumbel=g, mode="relaxed", temperature=parameters["temperature"])["weights"]
        return np.reshape(res, (-1, self.dim, self.n_classes))

    def kl_from_prior(self, parameters, eps=1e-20):
        if "logits" in parameters:
            logits = np.reshape(np.asarray(parameters["logits"]), (-1, self.n_classes))
        elif "probs" in parameters:
            q = np.reshape(np.asarray(parameters["probs"], dtype=np.float64), (-1, self.n_classes))
            logits = np.log(np.maximum(q, 1e-38))
        else:
            assert(False)
        z = np.zeros((self.n_classes, 1), np.float32)
        return _latent_eval(np.zeros((len(logits), 1), np.float32), np.zeros((len(logits), 1), np.float32), logits, z, z)["kl_c"]
