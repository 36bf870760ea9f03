"""Neural-filter backbone wrappers (mirror of the reference's src/models/ext/backbone.py).

``check_if_valid_target`` produces the image-level label the filter is trained on (:11-36);
``ExtIntermediateLayerGetter`` / ``ExtBackboneWithFPN`` run the same stem and layer engines as the distillation
path and add the reference's three exits: filter says "skip" (``None``), ``ext_training`` (stop after layer1),
or the full pyramid.
"""
from collections import OrderedDict

import torch

from ... import engine as E
from ... import hipnn
from ...hipnn import attach, to_nhwc
from ...myutils.pytorch import module_util
from ..mimic.base import BottleneckBase4Ext, ExtEncoder


def has_only_empty_bbox(target):
    return all(any(float(o) <= 1 for o in box[2:]) for box in target['boxes'])


def count_visible_keypoints(target):
    return sum(sum(1 for row in kp if float(row[2]) > 0) for kp in target['keypoints'])


def check_if_valid_target(target, min_keypoints_per_image=10):
    """no annotation / only degenerate boxes -> negative; keypoint targets need >= 10 visible joints (:19-36)."""
    if len(target) == 0 or has_only_empty_bbox(target):
        return False
    if 'keypoints' not in target:
        return True
    return count_visible_keypoints(target) >= min_keypoints_per_image


def check_if_includes_ext(module):
    return isinstance(module, BottleneckBase4Ext) and isinstance(module.encoder, ExtEncoder)


class ExtIntermediateLayerGetter(hipnn.IntermediateLayerGetter):
    def __init__(self, model, return_layers, ext_config):
        super().__init__(model, return_layers)
        self.threshold = ext_config['threshold']
        self.ext_training = False

    def get_ext_classifier(self):
        for module in self.values():
            if check_if_includes_ext(module):
                return module.get_ext_classifier()
        return None

    def forward(self, x):
        """(features dict | layer1 output when ext_training | None when the filter rejects, ext_z): reference :72-87"""
        x4 = to_nhwc(x, 4)
        if x4.shape[3] != 4:
            raise RuntimeError('the stem expects the 3-channel image batch stored as NHWC4')
        x0 = self.stem().forward(x4, False)             # the trunk is frozen while the filter is used / trained
        cur = attach(E.logical(x0), x0)
        out, ext_x = OrderedDict(), None
        for name, module in self.items():
            if name in ('conv1', 'bn1', 'relu', 'maxpool'):
                continue
            if isinstance(module, hipnn.ResLayer):
                module._keep = False
            cur = module(cur)
            if name in self.return_layers:
                if check_if_includes_ext(module) and isinstance(cur, tuple):
                    cur, ext_x = cur
                    if cur is None:
                        return None, ext_x
                    if self.ext_training:
                        return cur, ext_x
                out[self.return_layers[name]] = cur
        self._last_keep = False
        return out, ext_x


class ExtBackboneWithFPN(torch.nn.Module):
    def __init__(self, backbone, return_layers, in_channels_list, out_channels, ext_config):
        super().__init__()
        if ext_config.get('backbone_frozen', False):
            module_util.freeze_module_params(backbone)
        self.body = ExtIntermediateLayerGetter(backbone, return_layers=return_layers, ext_config=ext_config)
        self.fpn = hipnn.FeaturePyramidNetwork(in_channels_list, out_channels, extra_blocks=hipnn.LastLevelMaxPool())
        self.out_channels = out_channels
        self.split = False

    def forward(self, x):
        if self.split:
            raise NotImplementedError('head/tail split deployment is outside this build (SURVEY.md 8f-f1)')
        z, ext_z = self.body(x)
        if (not self.training and z is None) or self.body.ext_training:
            return None, ext_z
        if self.training:
            return z, ext_z
        return self.fpn(z), ext_z
