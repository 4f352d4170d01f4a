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
        """output widths of the FullyConnected layers, checking that consecutive ones chain"""
        w, prev = [], None
        for i, l in enumerate(self.layers):
            if not isinstance(l, FullyConnected):
                prev = None
                continue
            if prev is not None and l.input_dim != prev:
                raise ValueError("layer %d input_dim %d != previous output_dim %d" % (i + 1, l.input_dim, prev))
            w.append(l.output_dim)
            prev = l.output_dim
        return tuple(w)

    def conv_stack(self, side=28):
        """the convolutional front of the spec list as ((cin, cout, image side, pooled), ...): every Convolution with the
        MaxPooling that follows it folded in, image sides under SAME pooling (ceil(side / k))"""
        out = []
        for i, l in enumerate(self.layers):
            if isinstance(l, Convolution):
                if l.kernel != (3, 3) or l.strides != [1, 1, 1, 1]:
                    raise NotImplementedError("the step plan runs 3x3 stride-1 SAME convolutions (base_models.py:181-201)")
                out.append([l.prev_n_kernels, l.n_kernels, side, False])
            elif isinstance(l, MaxPooling):
                if l.k != 2 or not out or out[-1][3]:
                    raise NotImplementedError("the step plan runs one 2x2 SAME max-pool behind a convolution")
                out[-1][3] = True
                side = (side + 1) // 2
        return tuple(tuple(x) for x in out), side
