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
    """container that keeps the reference's ``encoder.encoder.N`` state-dict prefix; the optional neural-filter
    classifier (``ext_config``) belongs to ext_runner and is not built here."""

    def __init__(self, encoder, ext_classifier=None, ext_config=None):
        super().__init__()
        if ext_classifier is not None:
            raise NotImplementedError('neural filter (ext_config) belongs to ext_runner, outside this build')
        self.encoder, self.ext_classifier = encoder, None
        self.threshold = None if ext_config is None else ext_config['threshold']

    def get_ext_classifier(self):
        return self.ext_classifier

    def forward(self, x):
        raise RuntimeError('ExtEncoder executes fused inside Bottleneck4LargeResNet on the HIP path')


class BottleneckBase4Ext(nn.Module):
    """encoder -> [eval-only bottleneck transformer] -> decoder, executed as one HeadEngine plan."""

    def __init__(self, encoder, decoder, bottleneck_transformer=None):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.bottleneck_transformer = bottleneck_transformer
        self.use_bottleneck_transformer = False
        self.uses_ext_encoder = False
        self.data_logging = False
        self._engine = None

    def head_layers(self):
        """[(conv, pad, following BatchNorm2d, relu_after)] in execution order."""
        raise NotImplementedError

    def head_engine(self):
        if self._engine is None:
            self._engine = E.HeadEngine(self.head_layers())
        return self._engine

    def forward(self, x):
        use_codec = self.use_bottleneck_transformer and not self.training and self.bottleneck_transformer is not None
        out = self.head_engine().forward(to_nhwc(x), self.training,
                                         codec=self.bottleneck_transformer if use_codec else None)   # base.py:54-57
        return attach(E.logical(out), out)

    def get_ext_classifier(self):
        raise NotImplementedError('get_ext_classifier function is not implemented')
