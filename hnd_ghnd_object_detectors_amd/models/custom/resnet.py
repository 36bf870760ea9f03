"""Layer-injectable ResNet trunk (mirror of the reference's src/models/custom/resnet.py).

Same constructor surface and child order (conv1, bn1, relu, maxpool, layer1..4: the order
IntermediateLayerGetter relies on, reference :26-53), same initialisation (:55-60); the modules are
HIP-path holders/engines from ``hipnn`` instead of torch/torchvision layers.
"""
from torch import nn

from ... import hipnn


class CustomResNet(nn.Module):
    def __init__(self, block, layers, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, layer1=None, layer2=None, layer3=None,
                 layer4=None):
        super().__init__()
        if groups != 1 or width_per_group != 64 or (replace_stride_with_dilation and any(replace_stride_with_dilation)):
            raise NotImplementedError('HIP path: plain ResNet only (groups=1, width 64, no dilation)')
        self._norm_layer = norm_layer or hipnn.FrozenBatchNorm2d
        self.inplanes, self.dilation, self.groups, self.base_width = 64, 1, groups, width_per_group
        self.conv1 = hipnn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(64)
        self.relu = hipnn.ReLU(inplace=True)
        self.maxpool = hipnn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        injected = (layer1, layer2, layer3, layer4)
        for idx, (planes, nblocks, given) in enumerate(zip((64, 128, 256, 512), layers, injected), 1):
            if given is None:
                stage = self._make_layer(block, planes, nblocks, stride=1 if idx == 1 else 2)
            else:
                stage = given
                self.inplanes = planes * block.expansion
            setattr(self, 'layer%d' % idx, stage)
        for m in self.modules():                       # reference :55-60
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, hipnn.Bottleneck) and isinstance(m.bn3, nn.BatchNorm2d):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(hipnn.conv1x1(self.inplanes, planes * block.expansion, stride),
                                       self._norm_layer(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample, norm_layer=self._norm_layer)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return hipnn.ResLayer(*mods)

    def forward(self, x):
        raise RuntimeError('CustomResNet is executed through IntermediateLayerGetter (backbone.body) on the HIP path')


def _custom_resnet(arch, block, layers, pretrained, progress, strict=False, **kwargs):
    if pretrained:
        raise RuntimeError('ImageNet-pretrained %s weights must be supplied via a checkpoint: no network access' % arch)
    return CustomResNet(block, layers, **kwargs)


def custom_resnet50(pretrained=False, progress=True, **kwargs):
    return _custom_resnet('resnet50', hipnn.Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def custom_resnet101(pretrained=False, progress=True, **kwargs):
    return _custom_resnet('resnet101', hipnn.Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)


def custom_resnet152(pretrained=False, progress=True, **kwargs):
    return _custom_resnet('resnet152', hipnn.Bottleneck, [3, 8, 36, 3], pretrained, progress, **kwargs)


def resnet50(pretrained=False, progress=True, **kwargs):
    """torchvision.models.resnet.resnet50 equivalent (teacher trunk, reference rcnn.py:391)."""
    return _custom_resnet('resnet50', hipnn.Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet101(pretrained=False, progress=True, **kwargs):
    return _custom_resnet('resnet101', hipnn.Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)


def resnet152(pretrained=False, progress=True, **kwargs):
    return _custom_resnet('resnet152', hipnn.Bottleneck, [3, 8, 36, 3], pretrained, progress, **kwargs)
