"""Bottleneck-injected layer1 for ResNet-50/101/152 students (mirror of src/models/mimic/resnet_layer.py).

Module tree and state-dict keys follow the reference (:42-65): ``encoder.encoder.{0..7}`` and
``decoder.{0..11}``; eight bias-free 2x2 convs (encoder padded, decoder unpadded), eight train-mode
BatchNorm2d, four ReLU.  Execution is one fused HIP plan (engine.HeadEngine).
"""
from torch import nn

from ... import hipnn
from ..ext.classifier import Ext4ResNet
from .base import BottleneckBase4Ext, ExtEncoder


def _conv(cin, cout, padding):
    return hipnn.Conv2d(cin, cout, kernel_size=2, padding=padding, bias=False)


class Bottleneck4LargeResNet(BottleneckBase4Ext):
    def __init__(self, bottleneck_channel, ext_config, bottleneck_transformer):
        bn, relu = hipnn.BatchNorm2d, hipnn.ReLU
        encoder = nn.Sequential(
            _conv(64, 64, 1), bn(64),
            _conv(64, 256, 1), bn(256), relu(inplace=True),
            _conv(256, 64, 1), bn(64),
            _conv(64, bottleneck_channel, 1))
        decoder = nn.Sequential(
            bn(bottleneck_channel), relu(inplace=True),
            _conv(bottleneck_channel, 64, 0), bn(64),
            _conv(64, 128, 0), bn(128), relu(inplace=True),
            _conv(128, 256, 0), bn(256),
            _conv(256, 256, 0), bn(256), relu(inplace=True))
        ext_classifier = Ext4ResNet(64) if ext_config is not None else None          # reference :66
        super().__init__(encoder=ExtEncoder(encoder, ext_classifier, ext_config), decoder=decoder,
                         bottleneck_transformer=bottleneck_transformer)

    def head_layers(self):
        seq = list(self.encoder.encoder) + list(self.decoder)
        out = []
        for i, m in enumerate(seq):
            if isinstance(m, nn.Conv2d):
                bn = seq[i + 1]
                assert isinstance(bn, nn.BatchNorm2d)
                relu = i + 2 < len(seq) and isinstance(seq[i + 2], nn.ReLU)
                out.append((m, m.padding[0], bn, relu))
        assert len(out) == 8
        return out

    def get_ext_classifier(self):
        return self.encoder.get_ext_classifier()


# the reference builds the Large class for both names (resnet_layer.py:80-81)
Bottleneck4SmallResNet = Bottleneck4LargeResNet


def get_mimic_layers(backbone_name, backbone_config, bottleneck_transformer=None):
    layer1 = None
    layer1_config = backbone_config['params'].get('layer1', None)
    if layer1_config is not None:
        name = layer1_config['name']
        ext_config = backbone_config.get('ext_config', None)
        small = name == 'Bottleneck4SmallResNet' and backbone_name in {'custom_resnet18', 'custom_resnet34'}
        large = name == 'Bottleneck4LargeResNet' and backbone_name in {'custom_resnet50', 'custom_resnet101',
                                                                       'custom_resnet152'}
        if not (small or large):
            raise ValueError('layer1_name `{}` is not expected'.format(name))
        layer1 = Bottleneck4LargeResNet(layer1_config['bottleneck_channel'], ext_config, bottleneck_transformer)
    return layer1, None, None, None
