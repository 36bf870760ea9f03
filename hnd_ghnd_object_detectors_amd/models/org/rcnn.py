"""R-CNN detectors with the early-exit distillation forward (mirror of src/models/org/rcnn.py).

Public surface kept: ``CustomRCNNTransform``, ``CustomRCNN`` (``distill_backbone_only``, ``ext_training``),
``FasterRCNN`` / ``MaskRCNN`` / ``KeypointRCNN`` constructors with the reference's keyword arguments,
``MODEL_CLASS_DICT``, ``get_base_backbone``, ``get_fpn_backbone``, ``get_model``.  The transform and the
backbone run on the HIP engines; RPN / RoI heads are checkpoint-compatible holders (never executed
when ``distill_backbone_only`` is set, reference :109-110).
"""
import random
from collections import OrderedDict

import torch
from torch import nn

from ... import engine as E
from ... import hipnn
from .. import custom
from ..mimic.resnet_layer import get_mimic_layers


def resize_boxes(boxes, original_size, new_size):
    rh, rw = (float(s) / float(o) for s, o in zip(new_size, original_size))
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * rw, ymin * rh, xmax * rw, ymax * rh), dim=1)


def resize_masks_nearest(masks, scale):
    """ground-truth masks [k, h, w] uint8 -> [k, floor(h*s), floor(w*s)] uint8: the reference's
    ``misc_nn_ops.interpolate(mask[None].float(), scale_factor=scale)[0].byte()`` (src/models/org/rcnn.py:54-57, mode
    'nearest') as one HIP launch that moves bytes (hnd_resize_mask_nearest_u8) -- no float copy of the masks."""
    from ... import _lib, ops
    if not masks.is_cuda:
        raise RuntimeError('the HIP transform resizes ground-truth masks on the device: move the targets there first '
                           '(mimic_runner.distill_model does, as the reference :49-50)')
    m = masks.to(torch.uint8).contiguous()
    k, h, w = m.shape
    oh, ow = ops.interp_out_size(h, scale), ops.interp_out_size(w, scale)
    out = torch.empty((k, oh, ow), dtype=torch.uint8, device=m.device)
    _lib.check(_lib.load().hnd_resize_mask_nearest_u8(m.data_ptr(), k, h, w, oh, ow, float(scale), out.data_ptr(),
                                                      ops.stream_ptr()), 'hnd_resize_mask_nearest_u8')
    return out


def resize_keypoints(keypoints, original_size, new_size):
    rh, rw = (float(s) / float(o) for s, o in zip(new_size, original_size))
    out = keypoints.clone()
    out[..., 0] *= rw
    out[..., 1] *= rh
    return out


class CustomRCNNTransform(nn.Module):
    """normalise -> resize to a chosen min side (capped by max_size) -> zero-pad batch to /32 (reference :25-82).
    Image arithmetic is one HIP kernel per image; target boxes / masks / keypoints are rescaled in place like
    the reference (:50-62) with tiny torch ops (bookkeeping on a handful of numbers, not on the hot path)."""

    def __init__(self, min_size, max_size, image_mean, image_std):
        super().__init__()
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size, self.max_size = min_size, max_size
        self.image_mean, self.image_std = image_mean, image_std

    def choose_size(self, fixed_size=None):
        if fixed_size is not None:          # reference :34-40
            return fixed_size
        if self.training:
            return random.choice(self.min_size)
        return self.min_size[-1]

    def resize_target(self, target, old_hw, new_hw, scale, boxes_done=False):
        if target is None:
            return target
        if not boxes_done:
            target['boxes'] = resize_boxes(target['boxes'], old_hw, new_hw)
        if 'masks' in target:
            target['masks'] = resize_masks_nearest(target['masks'], scale)
        if 'keypoints' in target:
            target['keypoints'] = resize_keypoints(target['keypoints'], old_hw, new_hw)
        return target

    def forward(self, images, targets=None, fixed_sizes=None):
        images = list(images)
        sizes = [self.choose_size(fixed_sizes[i] if fixed_sizes is not None else None) for i in range(len(images))]
        eng = E.shared_transform(self.image_mean, self.image_std, images[0].device)
        batch, image_sizes = eng.run(images, sizes, self.max_size)
        if targets is not None:
            # the boxes of the whole batch in one HIP launch (the reference's per-image x * rw, y * rh in fp32, :50-53)
            idx = [i for i, t in enumerate(targets) if t is not None and t['boxes'].is_cuda
                   and t['boxes'].dtype == torch.float32 and t['boxes'].dim() == 2]
            if idx:
                from ... import ops
                items = []
                for i in idx:
                    oh, ow = tuple(images[i].shape[-2:])
                    nh, nw = image_sizes[i]
                    items.append((targets[i]['boxes'], float(nw) / float(ow), float(nh) / float(oh)))
                for i, out in zip(idx, ops.scale_boxes(items)):
                    targets[i]['boxes'] = out
            done = set(idx)
            for i, img in enumerate(images):
                targets[i] = self.resize_target(targets[i], tuple(img.shape[-2:]), image_sizes[i], eng.last_scales[i],
                                                boxes_done=i in done)
        tensors = hipnn.attach(E.logical(batch, 3), batch)
        return hipnn.ImageList(tensors, image_sizes), targets

    def postprocess(self, result, image_shapes, original_image_sizes):
        """GeneralizedRCNNTransform.postprocess: detections back to the original image frame -- boxes rescaled, masks
        pasted into full-size images (hnd_paste_masks), keypoints rescaled"""
        if self.training:
            return result
        from ...detection import resize_boxes as resize_back, paste_masks_in_image, resize_keypoints
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            boxes = resize_back(pred['boxes'], im_s, o_im_s)
            result[i]['boxes'] = boxes
            if 'masks' in pred:
                result[i]['masks'] = paste_masks_in_image(pred['masks'], boxes, o_im_s)
            if 'keypoints' in pred:
                result[i]['keypoints'] = resize_keypoints(pred['keypoints'], im_s, o_im_s)
        return result


class CustomRCNN(nn.Module):
    def __init__(self, backbone, rpn, roi_heads, transform):
        super().__init__()
        self.transform = transform
        self.backbone = backbone
        self.rpn = rpn
        self.roi_heads = roi_heads
        self.ext_training = False
        self.distill_backbone_only = False

    def train_ext(self):
        self.ext_training = True
        self.backbone.body.ext_training = True

    def get_ext_classifier(self):
        return self.backbone.body.get_ext_classifier()

    def forward(self, images, targets=None, fixed_sizes=None):
        if self.training and targets is None:
            raise ValueError('In training mode, targets should be passed')
        original_image_sizes = [tuple(img.shape[-2:]) for img in images]
        images, targets = self.transform(images, targets, fixed_sizes)
        features = self.backbone(images.tensors)
        if self.distill_backbone_only:
            return features
        if hasattr(self.backbone.body, 'ext_training'):         # ExtIntermediateLayerGetter, reference :113-122
            features, ext_logits = features
            if self.ext_training:
                return ext_logits
            if not self.training and features is None:          # the filter rejected the image: empty prediction
                ch, height, width = images.tensors.shape[1:]
                return [{'boxes': torch.empty(0, 4), 'labels': torch.empty(0, dtype=torch.int64),
                         'scores': torch.empty(0), 'masks': torch.zeros(100, ch, height, width),
                         'keypoints': torch.empty(0, 17, 3), 'keypoints_scores': torch.empty(0, 17)}]
        if self.training:
            raise NotImplementedError('detection losses (RPN / RoI heads in training mode) are never computed by the '
                                      'hnd/ghnd configs: they train with distill_backbone_only and org_loss_factor 0')
        if isinstance(features, torch.Tensor):
            features = OrderedDict([(0, features)])
        proposals, _ = self.rpn(images, features, targets)                      # reference :124-127
        detections, _ = self.roi_heads(features, proposals, images.image_sizes, targets)
        return self.transform.postprocess(detections, images.image_sizes, original_image_sizes)


class FasterRCNN(CustomRCNN):
    def __init__(self, backbone, num_classes=None, min_size=800, max_size=1333, image_mean=None, image_std=None,
                 rpn_anchor_generator=None, rpn_head=None, rpn_pre_nms_top_n_train=2000, rpn_pre_nms_top_n_test=1000,
                 rpn_post_nms_top_n_train=2000, rpn_post_nms_top_n_test=1000, rpn_nms_thresh=0.7,
                 rpn_fg_iou_thresh=0.7, rpn_bg_iou_thresh=0.3, rpn_batch_size_per_image=256,
                 rpn_positive_fraction=0.5, box_roi_pool=None, box_head=None, box_predictor=None,
                 box_score_thresh=0.05, box_nms_thresh=0.5, box_detections_per_img=100, box_fg_iou_thresh=0.5,
                 box_bg_iou_thresh=0.5, box_batch_size_per_image=512, box_positive_fraction=0.25,
                 bbox_reg_weights=None):
        if not hasattr(backbone, 'out_channels'):
            raise ValueError('backbone should contain an attribute out_channels specifying the number of output '
                             'channels (assumed to be the same for all the levels)')
        if num_classes is not None and box_predictor is not None:
            raise ValueError('num_classes should be None when box_predictor is specified')
        if num_classes is None and box_predictor is None:
            raise ValueError('num_classes should not be None when box_predictor is not specified')
        out_channels = backbone.out_channels
        if rpn_anchor_generator is None:
            rpn_anchor_generator = hipnn.AnchorGenerator(((32,), (64,), (128,), (256,), (512,)),
                                                         ((0.5, 1.0, 2.0),) * 5)
        if rpn_head is None:
            rpn_head = hipnn.RPNHead(out_channels, rpn_anchor_generator.num_anchors_per_location()[0])
        rpn = hipnn.RegionProposalNetwork(rpn_anchor_generator, rpn_head, rpn_fg_iou_thresh, rpn_bg_iou_thresh,
                                          rpn_batch_size_per_image, rpn_positive_fraction,
                                          dict(training=rpn_pre_nms_top_n_train, testing=rpn_pre_nms_top_n_test),
                                          dict(training=rpn_post_nms_top_n_train, testing=rpn_post_nms_top_n_test),
                                          rpn_nms_thresh)
        if box_roi_pool is None:
            box_roi_pool = hipnn.MultiScaleRoIAlign(featmap_names=[0, 1, 2, 3], output_size=7, sampling_ratio=2)
        if box_head is None:
            box_head = hipnn.TwoMLPHead(out_channels * box_roi_pool.output_size[0] ** 2, 1024)
        if box_predictor is None:
            box_predictor = hipnn.FastRCNNPredictor(1024, num_classes)
        roi_heads = hipnn.RoIHeads(box_roi_pool, box_head, box_predictor, box_fg_iou_thresh, box_bg_iou_thresh,
                                   box_batch_size_per_image, box_positive_fraction, bbox_reg_weights, box_score_thresh,
                                   box_nms_thresh, box_detections_per_img)
        transform = CustomRCNNTransform(min_size, max_size, image_mean or [0.485, 0.456, 0.406],
                                        image_std or [0.229, 0.224, 0.225])
        super().__init__(backbone, rpn, roi_heads, transform)


class MaskRCNN(FasterRCNN):
    def __init__(self, backbone, num_classes=None, mask_roi_pool=None, mask_head=None, mask_predictor=None, **kwargs):
        if num_classes is not None and mask_predictor is not None:
            raise ValueError('num_classes should be None when mask_predictor is specified')
        out_channels = backbone.out_channels
        super().__init__(backbone, num_classes, **kwargs)
        self.roi_heads.mask_roi_pool = mask_roi_pool or hipnn.MultiScaleRoIAlign([0, 1, 2, 3], 14, 2)
        self.roi_heads.mask_head = mask_head or hipnn.MaskRCNNHeads(out_channels, (256, 256, 256, 256), 1)
        self.roi_heads.mask_predictor = mask_predictor or hipnn.MaskRCNNPredictor(256, 256, num_classes)


class KeypointRCNN(FasterRCNN):
    def __init__(self, backbone, num_classes=None, min_size=None, keypoint_roi_pool=None, keypoint_head=None,
                 keypoint_predictor=None, num_keypoints=17, **kwargs):
        if min_size is None:
            min_size = (640, 672, 704, 736, 768, 800)
        if num_classes is not None and keypoint_predictor is not None:
            raise ValueError('num_classes should be None when keypoint_predictor is specified')
        out_channels = backbone.out_channels
        super().__init__(backbone, num_classes, min_size=min_size, **kwargs)
        self.roi_heads.keypoint_roi_pool = keypoint_roi_pool or hipnn.MultiScaleRoIAlign([0, 1, 2, 3], 14, 2)
        self.roi_heads.keypoint_head = keypoint_head or hipnn.KeypointRCNNHeads(out_channels, (512,) * 8)
        self.roi_heads.keypoint_predictor = keypoint_predictor or hipnn.KeypointRCNNPredictor(512, num_keypoints)


MODEL_URL_DICT = {
    'fasterrcnn_resnet50_fpn_coco': 'https://download.pytorch.org/models/fasterrcnn_resnet50_fpn_coco-258fb6c6.pth',
    'maskrcnn_resnet50_fpn_coco': 'https://download.pytorch.org/models/maskrcnn_resnet50_fpn_coco-bf2d0c1e.pth',
    'keypointrcnn_resnet50_fpn_coco': 'https://download.pytorch.org/models/keypointrcnn_resnet50_fpn_coco-fc266e95.pth',
}

MODEL_CLASS_DICT = {
    'faster_rcnn': (FasterRCNN, 'fasterrcnn_resnet50_fpn_coco'),
    'mask_rcnn': (MaskRCNN, 'maskrcnn_resnet50_fpn_coco'),
    'keypoint_rcnn': (KeypointRCNN, 'keypointrcnn_resnet50_fpn_coco'),
}


def get_base_backbone(backbone_name, backbone_config, bottleneck_transformer=None):
    pretrained = backbone_config['params']['pretrained']
    if backbone_name.startswith('resne'):
        return getattr(custom.resnet, backbone_name)(pretrained=pretrained, norm_layer=hipnn.FrozenBatchNorm2d)
    if backbone_name.startswith('custom_resne'):
        layer1, layer2, layer3, layer4 = get_mimic_layers(backbone_name, backbone_config, bottleneck_transformer)
        return getattr(custom.resnet, backbone_name)(pretrained=pretrained, norm_layer=hipnn.FrozenBatchNorm2d,
                                                     layer1=layer1, layer2=layer2, layer3=layer3, layer4=layer4)
    raise ValueError('backbone_name `{}` is not expected'.format(backbone_name))


def get_fpn_backbone(backbone, freeze_layers):
    if freeze_layers:                               # reference :400-403
        for name, parameter in backbone.named_parameters():
            if 'layer2' not in name and 'layer3' not in name and 'layer4' not in name:
                parameter.requires_grad_(False)
    return_layers = {'layer1': 0, 'layer2': 1, 'layer3': 2, 'layer4': 3}
    c2 = backbone.inplanes // 8
    return hipnn.BackboneWithFPN(backbone, return_layers, [c2, c2 * 2, c2 * 4, c2 * 8], 256)


def get_model_config(model_name):
    if model_name in MODEL_CLASS_DICT:
        return MODEL_CLASS_DICT[model_name]
    raise KeyError('model_name `{}` is not expected'.format(model_name))


def load_state_dict_from_url(url, progress=True):
    raise RuntimeError('cannot download %s (no network): point `ckpt` at a local checkpoint that holds the '
                       'torchvision COCO weights, or set params.pretrained: False' % url)


def get_model(model_name, pretrained, num_classes=91, backbone_config=None, custom_backbone=None, strict=True,
              progress=True, bottleneck_transformer=None, **kwargs):
    backbone_name = backbone_config['name']
    backbone_params_config = backbone_config['params']
    if pretrained:
        backbone_params_config['pretrained'] = False
    if custom_backbone is None:
        base_backbone = get_base_backbone(backbone_name, backbone_config, bottleneck_transformer)
        if backbone_config.get('ext_config', None) is not None:
            from ..ext import get_ext_fpn_backbone
            backbone = get_ext_fpn_backbone(base_backbone, backbone_config['ext_config'],
                                            backbone_params_config['freeze_layers'])
        else:
            backbone = get_fpn_backbone(base_backbone, backbone_params_config['freeze_layers'])
    else:
        backbone = custom_backbone
    model_class, pretrained_key = get_model_config(model_name)
    model = model_class(backbone, num_classes, **kwargs)
    if pretrained and backbone_name.endswith('resnet50'):
        print('Loading pretrained state dict of {}'.format(backbone_name))
        if backbone_name != 'resnet50':
            strict = False
        state_dict = load_state_dict_from_url(MODEL_URL_DICT[pretrained_key], progress=progress)
        model.load_state_dict(state_dict, strict=strict)
    return model
