"""Encoder/decoder container of the bottleneck-injected layer1 (mirror of src/models/mimic/base.py).

``BottleneckBase4Ext.forward`` (reference :50-58) = decoder(encoder(x)), with the eval-only bottleneck
transformer (:54-57: quantise / dequantise the bottleneck tensor) applied between the two when
``use_bottleneck_transformer`` is set.  With a neural filter (``ext_config``; :13-19, :38-48) the
layer returns ``(features or None, ext_z)``: the filter runs first on the stem output and, for batch-1 inference
below ``threshold``, the encoder/decoder are skipped (early exit).
"""
from torch import nn

from ... import engine as E
from ...hipnn import attach, to_nhwc


class ExtEncoder(nn.Module):
    """container that keeps the reference's ``encoder.encoder.N`` / ``encoder.ext_classifier.*`` state-dict
    prefixes; the convolutions execute inside the parent's HeadEngine, the classifier through its FilterEngine."""

    def __init__(self, encoder, ext_classifier=None, ext_config=None):
        super().__init__()
        self.encoder, self.ext_classifier = encoder, ext_classifier
        self.threshold = None if ext_config is None else ext_config['threshold']

    def get_ext_classifier(self):
        return self.ext_classifier

    def filter(self, x):
        """reference forward_with_ext (:13-19) up to the gate: (run the encoder?, ext_z)."""
        ext_z = self.ext_classifier(x)
        if not self.training and ext_z.shape[0] == 1 and float(ext_z[0][1]) < self.threshold:
            return False, ext_z
        return True, ext_z

    def forward(self, x):
        raise RuntimeError('ExtEncoder executes fused inside Bottleneck4LargeResNet on the HIP path')


class BottleneckBase4Ext(nn.Module):
    """encoder -> [eval-only bottleneck transformer] -> decoder, executed as one HeadEngine plan."""

    def __init__(self, encoder, decoder, bottleneck_transformer=None):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.bottleneck_transformer = bottleneck_transformer
        self.use_bottleneck_transformer = False
        self.uses_ext_encoder = isinstance(encoder, ExtEncoder) and encoder.ext_classifier is not None
        from ...structure.transformer import DataLogger
        self.data_logging = isinstance(bottleneck_transformer, DataLogger)       # reference :34
        self._engine = None

    def head_layers(self):
        """[(conv, pad, following BatchNorm2d, relu_after)] in execution order."""
        raise NotImplementedError

    def head_engine(self):
        if self._engine is None:
            self._engine = E.HeadEngine(self.head_layers())
        return self._engine

    def forward(self, x):
        ext_z = None
        if self.uses_ext_encoder:                       # reference :38-48 via ExtEncoder.forward_with_ext
            go_on, ext_z = self.encoder.filter(x)
            if not go_on:
                if self.data_logging:                   # reference :40-42: a rejected image logs an empty bottleneck
                    self.bottleneck_transformer(None, target=None)
                return None, ext_z
        use_codec = self.use_bottleneck_transformer and not self.training and self.bottleneck_transformer is not None
        eng = self.head_engine()
        out = eng.forward(to_nhwc(x), self.training,
                          codec=self.bottleneck_transformer if use_codec else None)   # base.py:54-57
        body = self.__dict__.get('_body')
        z = None
        if self.encoder._forward_hooks or self.decoder._forward_hooks:
            # hooks on ``...layer1.encoder`` / ``...layer1.decoder`` (src/distillation/tool.py:22-35 takes any path): the
            # encoder's output is the bottleneck tensor z (raw output of its last conv; the BatchNorm that follows is
            # decoder.0), the decoder's output is the layer's
            from ...hipnn import fire_forward_hooks
            zi = eng.encoder_len - 1
            zbuf = eng.y[zi]
            z = attach(E.logical(zbuf, eng.layers[zi].cout), zbuf, (body, 'layer1', 'encoder'))
            fire_forward_hooks(self.encoder, x, z)
            fire_forward_hooks(self.decoder, z, attach(E.logical(out), out, (body, 'layer1', 'decoder')))
        out = attach(E.logical(out), out)
        return (out, ext_z) if self.uses_ext_encoder else out

    def get_ext_classifier(self):
        raise NotImplementedError('get_ext_classifier function is not implemented')
