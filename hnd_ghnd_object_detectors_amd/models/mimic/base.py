"""Encoder/decoder container of the bottleneck-injected layer1 (mirror of src/models/mimic/base.py).

``BottleneckBase4Ext.forward`` (reference :50-58) = decoder(encoder(x)), with the eval-only bottleneck
transformer (:54-57: quantise / dequantise the bottleneck tensor) applied between the two when
``use_bottleneck_transformer`` is set; the neural-filter branch (:13-19, :38-48) belongs to ext_runner
and raises.
"""
from torch import nn

from ... import engine as E
from ...hipnn import attach, to_nhwc


class ExtEncoder(nn.Module):
    def __init__(self, encoder, ext_classifier=None, ext_config=None):
        super().__init__()
        self.encoder = encoder
        self.ext_classifier = ext_classifier
        self.threshold = ext_config['threshold'] if ext_config is not None else None
        if ext_classifier is not None:
            raise NotImplementedError('neural filter (ext_config) belongs to ext_runner, outside this build')

    def forward(self, x):
        raise RuntimeError('ExtEncoder executes fused inside Bottleneck4LargeResNet on the HIP path')

    def get_ext_classifier(self):
        return self.ext_classifier


class BottleneckBase4Ext(nn.Module):
    def __init__(self, encoder, decoder, bottleneck_transformer=None):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.bottleneck_transformer = bottleneck_transformer
        self.data_logging = False
        self.uses_ext_encoder = isinstance(encoder, ExtEncoder) and encoder.ext_classifier is not None
        self.use_bottleneck_transformer = False
        self._engine = None

    def head_layers(self):
        """[(conv, pad, following BatchNorm2d, relu_after)] in execution order."""
        raise NotImplementedError

    def head_engine(self):
        if self._engine is None:
            self._engine = E.HeadEngine(self.head_layers())
        return self._engine

    def forward(self, x):
        codec = None
        if (not self.training) and self.bottleneck_transformer is not None and self.use_bottleneck_transformer:
            codec = self.bottleneck_transformer            # reference base.py:54-57 (eval only)
        eng = self.head_engine()
        out = eng.forward(to_nhwc(x), self.training, codec=codec)
        return attach(E.logical(out), out)

    def get_ext_classifier(self):
        raise NotImplementedError('get_ext_classifier function is not implemented')
