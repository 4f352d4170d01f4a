"""Layer descriptors (code/includes/layers.py).  In the reference these classes
create TensorFlow variables and ops; here they only describe the architecture
that the HIP step plan executes (dense + bias + ReLU = one GEMM with a fused
epilogue).  Convolution / MaxPooling / BatchNormalization belong to the CNN
trunk, a "next" row of SURVEY.md 8f."""


class Layer:
    def __init__(self, name, activation="relu", initializer="xavier"):
        self.name = name
        self.activation = activation
        self.initializer = initializer


class FullyConnected(Layer):
    """relu(x W + b), W (in,out) and b (1,out) both xavier-initialised
    (includes/layers.py:19-36)."""

    def __init__(self, name, input_dim, output_dim, activation="relu", initializer="xavier"):
        Layer.__init__(self, name, activation=activation, initializer=initializer)
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)


def _unsupported(kind):
    class _U(Layer):
        def __init__(self, *a, **k):
            raise NotImplementedError("%s: the CNN encoder trunk is not part of the MLP hot path (SURVEY.md F1, 8f)" % kind)
    _U.__name__ = kind
    return _U


Convolution = _unsupported("Convolution")
MaxPooling = _unsupported("MaxPooling")
BatchNormalization = _unsupported("BatchNormalization")
