"""Neural-filter backbone wrappers (role of the reference's src/models/ext/backbone.py).

``check_if_valid_target`` is the image-level label the filter is trained on (reference :11-36).
``ExtIntermediateLayerGetter`` / ``ExtBackboneWithFPN`` run the same stem and layer engines as the distillation
path and add the reference's three exits (:72-87, :107-116): the filter rejects the image (``None``),
``ext_training`` (stop after layer1), or the full pyramid.
"""
from collections import OrderedDict

import torch

from ... import engine as E
from ... import hipnn
from ...hipnn import attach, to_nhwc
from ...myutils.pytorch import module_util
from ..mimic.base import BottleneckBase4Ext, ExtEncoder

MIN_KEYPOINTS_PER_IMAGE = 10
_STEM_CHILDREN = ('conv1', 'bn1', 'relu', 'maxpool')


def has_only_empty_bbox(target):
    """every box is degenerate: width or height entry <= 1 (the rule reads the last two numbers of each box)"""
    for box in target['boxes']:
        if not any(float(v) <= 1 for v in box[2:]):
            return False
    return True


def count_visible_keypoints(target):
    return int(sum(int((kp[:, 2] > 0).sum()) for kp in target['keypoints']))


def check_if_valid_target(target, min_keypoints_per_image=MIN_KEYPOINTS_PER_IMAGE):
    """positive label <=> annotated, not only degenerate boxes and, for keypoint targets, enough visible joints"""
    annotated = len(target) > 0 and not has_only_empty_bbox(target)
    if not annotated or 'keypoints' not in target:
        return annotated
    return count_visible_keypoints(target) >= min_keypoints_per_image


def check_if_includes_ext(module):
    return isinstance(module, BottleneckBase4Ext) and isinstance(module.encoder, ExtEncoder)


class ExtIntermediateLayerGetter(hipnn.IntermediateLayerGetter):
    def __init__(self, model, return_layers, ext_config):
        super().__init__(model, return_layers)
        self.threshold, self.ext_training = ext_config['threshold'], False

    def get_ext_classifier(self):
        holders = [m for m in self.values() if check_if_includes_ext(m)]
        return holders[0].get_ext_classifier() if holders else None

    def forward(self, x):
        """-> (features | layer1 output when ext_training | None when rejected, filter output)"""
        batch = to_nhwc(x, 4)
        if batch.shape[3] != 4:
            raise RuntimeError('the stem expects the 3-channel image batch stored as NHWC4')
        x0 = self.stem().forward(batch, False)          # the trunk is frozen while the filter is used / trained
        cur, ext_x, feats = attach(E.logical(x0), x0), None, OrderedDict()
        self._last_keep = False
        for name, module in self.items():
            if name in _STEM_CHILDREN:
                continue
            if isinstance(module, hipnn.ResLayer):
                module._keep = False
            cur = module(cur)
            if name not in self.return_layers:
                continue
            if isinstance(cur, tuple):                  # the bottleneck layer with a filter: (features, ext_z)
                cur, ext_x = cur
                if cur is None or self.ext_training:
                    return cur, ext_x
            feats[self.return_layers[name]] = cur
        return feats, ext_x


class ExtBackboneWithFPN(torch.nn.Module):
    def __init__(self, backbone, return_layers, in_channels_list, out_channels, ext_config):
        super().__init__()
        if ext_config.get('backbone_frozen', False):
            module_util.freeze_module_params(backbone)
        self.body = ExtIntermediateLayerGetter(backbone, return_layers=return_layers, ext_config=ext_config)
        self.fpn = hipnn.FeaturePyramidNetwork(in_channels_list, out_channels, extra_blocks=hipnn.LastLevelMaxPool())
        self.out_channels, self.split = out_channels, False

    def forward(self, x):
        if self.split:
            raise NotImplementedError('use models.mimic.split_rcnn.split_rcnn_model for the head/tail deployment')
        z, ext_z = self.body(x)
        rejected = z is None and not self.training
        if rejected or self.body.ext_training:
            return None, ext_z
        return (z if self.training else self.fpn(z)), ext_z
