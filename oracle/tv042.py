"""Restatement of the torchvision==0.4.2 symbols used by the reference path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torchvision 0.4.2 is pinned by the
reference (``Pipfile:8``) but is absent from this image, so its *published*
behaviour is restated here in plain torch.  Call sites in the reference:
``src/models/org/rcnn.py:6-17,391,394,414,228`` and ``src/models/custom/resnet.py:2-3``.

Only the pieces that EXECUTE on the distillation step are behaviourally faithful
(ResNet Bottleneck, FrozenBatchNorm2d without eps, IntermediateLayerGetter, FPN,
GeneralizedRCNNTransform.normalize/batch_images, ImageList).  The eval-mode RPN / RoI heads
(box, mask and keypoint branches) of the validation path are restated in
oracle/tv042_det.py; the mask / keypoint head modules below are plain nn.Sequential / nn.Module
stacks with the 0.4.2 names and shapes.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

model_urls = {
    'resnet18': 'https://download.pytorch.org/models/resnet18-5c106cde.pth',
    'resnet34': 'https://download.pytorch.org/models/resnet34-333f7ec4.pth',
    'resnet50': 'https://download.pytorch.org/models/resnet50-19c8e357.pth',
    'resnet101': 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth',
    'resnet152': 'https://download.pytorch.org/models/resnet152-b121ed2d.pth',
}


def load_state_dict_from_url(url, progress=True, **kwargs):
    raise RuntimeError('no network in this environment: cannot fetch %s' % url)


# --------------------------------------------------------------------------- resnet
def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation,
                     groups=groups, bias=False, dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1,
                 base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + shortcut)


class Bottleneck(nn.Module):
    """v1.5 bottleneck: the stride sits on the 3x3 ``conv2``."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1,
                 base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += shortcut
        return self.relu(y)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self.inplanes, self.dilation = 64, 1
        dil = replace_stride_with_dilation or [False, False, False]
        self.groups, self.base_width = groups, width_per_group
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=dil[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=dil[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=dil[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer, downsample, prev_dil = self._norm_layer, None, self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width,
                      prev_dil, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            mods.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                              dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def _resnet(arch, block, layers, pretrained, progress, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        model.load_state_dict(load_state_dict_from_url(model_urls[arch], progress=progress))
    return model


def resnet18(pretrained=False, progress=True, **kw):
    return _resnet('resnet18', BasicBlock, [2, 2, 2, 2], pretrained, progress, **kw)


def resnet34(pretrained=False, progress=True, **kw):
    return _resnet('resnet34', BasicBlock, [3, 4, 6, 3], pretrained, progress, **kw)


def resnet50(pretrained=False, progress=True, **kw):
    return _resnet('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, progress, **kw)


def resnet101(pretrained=False, progress=True, **kw):
    return _resnet('resnet101', Bottleneck, [3, 4, 23, 3], pretrained, progress, **kw)


def resnet152(pretrained=False, progress=True, **kw):
    return _resnet('resnet152', Bottleneck, [3, 8, 36, 3], pretrained, progress, **kw)


# --------------------------------------------------------------------------- ops.misc
class FrozenBatchNorm2d(nn.Module):
    """0.4.2 semantics: per-channel affine from buffers, NO eps, no num_batches_tracked."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))

    def forward(self, x):
        scale = self.weight * self.running_var.rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def interpolate(input, size=None, scale_factor=None, mode='nearest', align_corners=None):
    return F.interpolate(input, size, scale_factor, mode, align_corners)


class MiscConv2d(nn.Conv2d):
    pass


class MiscConvTranspose2d(nn.ConvTranspose2d):
    pass


# --------------------------------------------------------------------------- models._utils
class IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError('return_layers are not present in model')
        orig = return_layers
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.named_children():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


# --------------------------------------------------------------------------- FPN
class LastLevelMaxPool(nn.Module):
    def forward(self, x, names):
        names.append('pool')
        x.append(F.max_pool2d(x[-1], 1, 2, 0))
        return x, names


class FeaturePyramidNetwork(nn.Module):
    def __init__(self, in_channels_list, out_channels, extra_blocks=None):
        super().__init__()
        self.inner_blocks = nn.ModuleList()
        self.layer_blocks = nn.ModuleList()
        for c in in_channels_list:
            if c == 0:
                continue
            self.inner_blocks.append(nn.Conv2d(c, out_channels, 1))
            self.layer_blocks.append(nn.Conv2d(out_channels, out_channels, 3, padding=1))
        # 0.4.2 iterates children() (ModuleLists), so this init never fires; kept for fidelity
        for m in self.children():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
        self.extra_blocks = extra_blocks

    def forward(self, x):
        names, feats = list(x.keys()), list(x.values())
        last_inner = self.inner_blocks[-1](feats[-1])
        results = [self.layer_blocks[-1](last_inner)]
        for i in range(len(feats) - 2, -1, -1):
            lateral = self.inner_blocks[i](feats[i])
            top_down = F.interpolate(last_inner, size=lateral.shape[-2:], mode='nearest')
            last_inner = lateral + top_down
            results.insert(0, self.layer_blocks[i](last_inner))
        if self.extra_blocks is not None:
            results, names = self.extra_blocks(results, names)
        return OrderedDict(zip(names, results))


class BackboneWithFPN(nn.Sequential):
    def __init__(self, backbone, return_layers, in_channels_list, out_channels):
        body = IntermediateLayerGetter(backbone, return_layers=return_layers)
        fpn = FeaturePyramidNetwork(in_channels_list, out_channels, extra_blocks=LastLevelMaxPool())
        super().__init__(OrderedDict([('body', body), ('fpn', fpn)]))
        self.out_channels = out_channels


# --------------------------------------------------------------------------- detection.transform
class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def resize_boxes(boxes, original_size, new_size):
    rh, rw = (float(s) / float(o) for s, o in zip(new_size, original_size))
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * rw, ymin * rh, xmax * rw, ymax * rh), dim=1)


def resize_keypoints(keypoints, original_size, new_size):
    rh, rw = (float(s) / float(o) for s, o in zip(new_size, original_size))
    out = keypoints.clone()
    out[..., 0] *= rw
    out[..., 1] *= rh
    return out


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size, max_size, image_mean, image_std):
        super().__init__()
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size, self.max_size = min_size, max_size
        self.image_mean, self.image_std = image_mean, image_std

    def normalize(self, image):
        mean = torch.as_tensor(self.image_mean, dtype=image.dtype, device=image.device)
        std = torch.as_tensor(self.image_std, dtype=image.dtype, device=image.device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def batch_images(self, images, size_divisible=32):
        max_size = [max(s) for s in zip(*[img.shape for img in images])]
        max_size[1] = int(math.ceil(float(max_size[1]) / size_divisible) * size_divisible)
        max_size[2] = int(math.ceil(float(max_size[2]) / size_divisible) * size_divisible)
        batched = images[0].new_zeros((len(images),) + tuple(max_size))
        for img, pad in zip(images, batched):
            pad[:img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        return batched

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            boxes = resize_boxes(pred['boxes'], im_s, o_im_s)
            result[i]['boxes'] = boxes
            if 'masks' in pred:
                result[i]['masks'] = paste_masks_in_image(pred['masks'], boxes, o_im_s)
            if 'keypoints' in pred:
                result[i]['keypoints'] = resize_keypoints(pred['keypoints'], im_s, o_im_s)
        return result


# --------------------------------------------------------------------------- detection heads
# RPN / RoI heads (eval mode, box + mask + keypoint branches): restatements live in oracle/tv042_det.py (row f4)
from oracle.tv042_det import (AnchorGenerator, RPNHead, RegionProposalNetwork, concat_box_prediction_layers,  # noqa: E402,F401
                              MultiScaleRoIAlign, TwoMLPHead, FastRCNNPredictor, RoIHeads, BoxCoder,
                              paste_masks_in_image)


def _not_on_path(name):
    def forward(self, *a, **k):
        raise NotImplementedError('%s is outside the restated paths (oracle holder only)' % name)
    return forward


class MaskRCNNHeads(nn.Sequential):
    def __init__(self, in_channels, layers, dilation):
        d, nxt = OrderedDict(), in_channels
        for i, feat in enumerate(layers, 1):
            d['mask_fcn{}'.format(i)] = MiscConv2d(nxt, feat, kernel_size=3, stride=1,
                                                   padding=dilation, dilation=dilation)
            d['relu{}'.format(i)] = nn.ReLU(inplace=True)
            nxt = feat
        super().__init__(d)


class MaskRCNNPredictor(nn.Sequential):
    def __init__(self, in_channels, dim_reduced, num_classes):
        super().__init__(OrderedDict([
            ('conv5_mask', MiscConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)),
            ('relu', nn.ReLU(inplace=True)),
            ('mask_fcn_logits', MiscConv2d(dim_reduced, num_classes, 1, 1, 0))]))


class KeypointRCNNHeads(nn.Sequential):
    def __init__(self, in_channels, layers):
        d, nxt = [], in_channels
        for feat in layers:
            d.append(MiscConv2d(nxt, feat, 3, stride=1, padding=1))
            d.append(nn.ReLU(inplace=True))
            nxt = feat
        super().__init__(*d)


class KeypointRCNNPredictor(nn.Module):
    def __init__(self, in_channels, num_keypoints):
        super().__init__()
        self.kps_score_lowres = MiscConvTranspose2d(in_channels, num_keypoints, 4, stride=2, padding=1)
        self.up_scale = 2
        self.out_channels = num_keypoints

    def forward(self, x):
        x = self.kps_score_lowres(x)
        return F.interpolate(x, scale_factor=self.up_scale, mode='bilinear', align_corners=False)


class _DetectionBase(nn.Module):
    """Stand-ins for torchvision.models.detection.{FasterRCNN,MaskRCNN,KeypointRCNN}: the
    reference only uses them in ``isinstance`` checks (``src/models/__init__.py:66-68``)."""


class TVFasterRCNN(_DetectionBase):
    pass


class TVMaskRCNN(TVFasterRCNN):
    pass


class TVKeypointRCNN(TVFasterRCNN):
    pass
