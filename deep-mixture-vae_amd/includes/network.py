"""Sequential container of layer specs (code/includes/network.py:57-82)."""
from includes.layers import FullyConnected, Convolution, MaxPooling, BatchNormalization

_layers_id_mapping = {"fc": FullyConnected, "cn": Convolution, "mp": MaxPooling, "bn": BatchNormalization}


class DeepNetwork:
    def __init__(self, name, layers, activation="relu", initializer="xavier"):
        self.name = name
        self.layers = []
        for index, (layer_id, args) in enumerate(layers):
            if layer_id not in _layers_id_mapping:
                raise NotImplementedError
            self.layers.append(_layers_id_mapping[layer_id]("layer_%d" % (index + 1), activation=activation,
                                                            initializer=initializer, **args))

    def widths(self):
        w = []
        for i, l in enumerate(self.layers):
            if i and l.input_dim != self.layers[i - 1].output_dim:
                raise ValueError("layer %d input_dim %d != previous output_dim %d" % (i + 1, l.input_dim, self.layers[i - 1].output_dim))
            w.append(l.output_dim)
        return tuple(w)
