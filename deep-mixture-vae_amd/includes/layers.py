"""Layer descriptors (code/includes/layers.py).  In the reference these classes create TensorFlow variables and ops;
here they DESCRIBE the architecture that the HIP step plan executes -- dense + bias + ReLU is one GEMM with a fused
epilogue, a 3x3 SAME convolution + bias + ReLU is one conv-mode GEMM, a 2x2 SAME max-pool is one kernel
(csrc/conv.hip) -- and carry the shapes the reference gives its variables, so that a DeepNetwork spec list written
as in code/base_models.py:179-216 / :280-288 drives the plan."""


class Layer:
    def __init__(self, name, activation="relu", initializer="xavier"):
        self.name = name
        self.activation = activation
        self.initializer = initializer


class FullyConnected(Layer):
    """relu(flatten(x) W + b), W (in, out) and b (1, out) both xavier-initialised (includes/layers.py:19-36)."""

    def __init__(self, name, input_dim, output_dim, activation="relu", initializer="xavier"):
        Layer.__init__(self, name, activation=activation, initializer=initializer)
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.weight_shape, self.bias_shape = (self.input_dim, self.output_dim), (1, self.output_dim)


class Convolution(Layer):
    """relu(bias_add(conv2d(x, W, strides, 'SAME'), b)), W (kh, kw, prev_n_kernels, n_kernels) and b (n_kernels,) both from
    the initializer (includes/layers.py:39-60).  The step plan runs 3x3 kernels at stride 1 (what base_models.py:181-201 uses)."""

    def __init__(self, name, n_kernels, prev_n_kernels, kernel, strides=1, activation="relu", initializer="xavier"):
        Layer.__init__(self, name, activation=activation, initializer=initializer)
        self.n_kernels, self.prev_n_kernels = int(n_kernels), int(prev_n_kernels)
        self.kernel = tuple(int(v) for v in kernel)
        self.strides = [1, int(strides), int(strides), 1]
        self.weight_shape, self.bias_shape = self.kernel + (self.prev_n_kernels, self.n_kernels), (self.n_kernels,)


class MaxPooling(Layer):
    """max_pool(x, ksize k, strides k, 'SAME') (includes/layers.py:63-77); the plan runs k = 2."""

    def __init__(self, name, k, activation=None, initializer=None):
        Layer.__init__(self, name, activation=activation, initializer=initializer)
        self.k = int(k)
        self.ksize = [1, self.k, self.k, 1]
        self.strides = [1, self.k, self.k, 1]


class BatchNormalization(Layer):
    """includes/layers.py:80-92.  No model of the reference instantiates it and the plan has no kernel for it."""

    def __init__(self, name, is_training, activation=None, initializer=None):
        raise NotImplementedError("BatchNormalization: never instantiated by the reference's models; not part of the step plan")
