"""Restatement of torchvision==0.4.2's EVAL-MODE detection machinery.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference reaches this code from
``src/models/org/rcnn.py:124-127`` (``self.rpn`` -> ``self.roi_heads`` -> ``transform.postprocess``) whenever
``distill_backbone_only`` is False, i.e. on the validation path ``src/utils/main_util.py:75-113`` that selects the
checkpoint (``src/mimic_runner.py:94-100``).  torchvision 0.4.2 (``Pipfile:8``) is absent from the image, so the
*published* algorithm of these files is restated in plain CPU torch:

    torchvision/models/detection/rpn.py         AnchorGenerator, RPNHead, RegionProposalNetwork (eval branch)
    torchvision/models/detection/_utils.py      BoxCoder.decode
    torchvision/models/detection/roi_heads.py   RoIHeads.forward / postprocess_detections (eval), maskrcnn_inference,
                                                paste_masks_in_image, keypointrcnn_inference, heatmaps_to_keypoints
    torchvision/ops/boxes.py                    nms, batched_nms, clip_boxes_to_image, remove_small_boxes, box_area
    torchvision/ops/poolers.py                  MultiScaleRoIAlign, LevelMapper
    torchvision/csrc/cpu/{nms_cpu,ROIAlign_cpu}.cpp   the two native CPU operators

PARITY UNPINNED for the third-party part: no torchvision 0.4.2 binary exists here to run, so version-sensitive
details are restated from the 0.4.2 sources as published and flagged below:
  * ``nms`` (CPU operator) returns the kept indices in ASCENDING INDEX order (``at::nonzero(suppressed == 0)``);
    score order came with 0.5.  Callers slice ``keep[:n]`` on that order (rpn.py, roi_heads.py).
  * ``nms`` TIE RULE: a box is suppressed when ``ovr > iou_threshold`` (STRICT), as 0.4.2's ``nms_cpu_kernel``
    (``if (ovr > iou_threshold) suppressed_t[j] = 1;``).  The maskrcnn-benchmark kernel it was ported from used ``>=``;
    the two differ only on boxes whose IoU equals the threshold exactly (measure zero on real detections, reachable
    with the duplicated / quantised boxes of the tests).  Strict is taken here AND in ``csrc/detect.hip``
    (``if (ovr > thr)``): tests/test_detect_gpu.py holds the two to identical kept sets including exact-threshold ties.
  * anchor strides are the true quotients ``image_size / grid_size`` (floats; 800/13 for the 'pool' level).
  * ``postprocess_detections`` has no "remove empty boxes" step (added in 0.5).
  * ``roi_align`` is the non-"aligned" form (no half-pixel shift; roi width/height clamped to >= 1).
  * ``paste_masks_in_image``: masks zero-padded by 1 px, boxes expanded by (M+2)/M about their centre and truncated
    to int64, each mask resized to (int(y1-y0+1), int(x1-x0+1)) with bilinear ``interpolate(align_corners=False)`` and
    pasted into the clipped window (the 0.4.x Python loop; the sampled-grid GPU paste came with 0.6).
  * ``heatmaps_to_keypoints``: per RoI BICUBIC ``interpolate(align_corners=False)`` of the 56x56 heatmaps to
    (ceil(h), ceil(w)), argmax per keypoint, ``(idx + 0.5) * size / ceil(size) + offset``; visibility column = 1;
    ``y_int = (pos - x_int) / w`` is an exact integer quotient.
What IS pinned: the reference's own ``rcnn.py`` forward runs UNMODIFIED over these classes when
tests/golden/make_golden.py writes the ``tiny_detect_*`` fixtures, and the HIP path is tested against them.
Training-mode branches (RPN / RoI losses, matchers, samplers) are not restated: every hnd/ghnd config trains with
``org_loss_factor: 0`` and ``distill_backbone_only`` (they never run).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------------- ops/boxes.py
def box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def clip_boxes_to_image(boxes, size):
    dim = boxes.dim()
    boxes_x = boxes[..., 0::2]
    boxes_y = boxes[..., 1::2]
    height, width = size
    boxes_x = boxes_x.clamp(min=0, max=width)
    boxes_y = boxes_y.clamp(min=0, max=height)
    clipped = torch.stack((boxes_x, boxes_y), dim=dim)
    return clipped.reshape(boxes.shape)


def remove_small_boxes(boxes, min_size):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    keep = (ws >= min_size) & (hs >= min_size)
    return keep.nonzero().squeeze(1)


def nms(boxes, scores, iou_threshold):
    """torchvision/csrc/cpu/nms_cpu.cpp (0.4.2): greedy suppression in descending-score order; returns
    ``nonzero(suppressed == 0)`` = the kept indices in ascending index order."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    x1, y1, x2, y2 = boxes[:, 0].contiguous(), boxes[:, 1].contiguous(), boxes[:, 2].contiguous(), boxes[:, 3].contiguous()
    areas = (x2 - x1) * (y2 - y1)
    order = scores.sort(0, descending=True)[1]
    n = boxes.shape[0]
    suppressed = torch.zeros(n, dtype=torch.bool)
    zero = boxes.new_zeros(())
    for _i in range(n):
        i = int(order[_i])
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        if rest.numel() == 0:
            break
        xx1 = torch.max(x1[i], x1[rest])
        yy1 = torch.max(y1[i], y1[rest])
        xx2 = torch.min(x2[i], x2[rest])
        yy2 = torch.min(y2[i], y2[rest])
        w = torch.max(zero, xx2 - xx1)
        h = torch.max(zero, yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > iou_threshold]] = True        # already-suppressed entries stay suppressed
    return (~suppressed).nonzero().squeeze(1)


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    boxes_for_nms = boxes + offsets[:, None]
    return nms(boxes_for_nms, scores, iou_threshold)


# --------------------------------------------------------------------------------------------- ROIAlign_cpu.cpp
def roi_align(input, rois, output_size, spatial_scale, sampling_ratio):
    """input [N, C, H, W]; rois [K, 5] = (batch index, x1, y1, x2, y2) in image coordinates.
    Average of ``sampling_ratio^2`` bilinear samples per bin; a sample outside [-1, H] x [-1, W] contributes 0."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    k = rois.shape[0]
    n, c, height, width = input.shape
    out = input.new_zeros((k, c, ph, pw))
    if k == 0:
        return out
    one = input.new_ones(())
    for r in range(k):
        b = int(rois[r, 0])
        roi_start_w, roi_start_h = rois[r, 1] * spatial_scale, rois[r, 2] * spatial_scale
        roi_end_w, roi_end_h = rois[r, 3] * spatial_scale, rois[r, 4] * spatial_scale
        roi_w = torch.max(roi_end_w - roi_start_w, one)
        roi_h = torch.max(roi_end_h - roi_start_h, one)
        bin_h, bin_w = roi_h / ph, roi_w / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(roi_h) / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(roi_w) / pw))
        count = gh * gw
        # sample coordinates: y[ph][iy], x[pw][ix]
        iy = (torch.arange(gh, dtype=input.dtype) + 0.5)
        ix = (torch.arange(gw, dtype=input.dtype) + 0.5)
        ys = roi_start_h + torch.arange(ph, dtype=input.dtype)[:, None] * bin_h + iy[None, :] * bin_h / gh
        xs = roi_start_w + torch.arange(pw, dtype=input.dtype)[:, None] * bin_w + ix[None, :] * bin_w / gw
        ys, xs = ys.reshape(-1), xs.reshape(-1)                                  # [ph*gh], [pw*gw]
        y_bad, x_bad = (ys < -1.0) | (ys > height), (xs < -1.0) | (xs > width)
        yc, xc = ys.clamp(min=0), xs.clamp(min=0)
        y_low, x_low = yc.long(), xc.long()
        y_edge, x_edge = y_low >= height - 1, x_low >= width - 1
        y_low = torch.where(y_edge, torch.full_like(y_low, height - 1), y_low)
        x_low = torch.where(x_edge, torch.full_like(x_low, width - 1), x_low)
        y_high = torch.where(y_edge, y_low, y_low + 1)
        x_high = torch.where(x_edge, x_low, x_low + 1)
        yc = torch.where(y_edge, y_low.to(input.dtype), yc)
        xc = torch.where(x_edge, x_low.to(input.dtype), xc)
        ly, lx = yc - y_low.to(input.dtype), xc - x_low.to(input.dtype)
        hy, hx = 1.0 - ly, 1.0 - lx
        feat = input[b]                                                           # [C, H, W]
        v1 = feat[:, y_low][:, :, x_low]
        v2 = feat[:, y_low][:, :, x_high]
        v3 = feat[:, y_high][:, :, x_low]
        v4 = feat[:, y_high][:, :, x_high]
        w1, w2 = hy[:, None] * hx[None, :], hy[:, None] * lx[None, :]
        w3, w4 = ly[:, None] * hx[None, :], ly[:, None] * lx[None, :]
        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4                               # [C, ph*gh, pw*gw]
        val = torch.where((y_bad[:, None] | x_bad[None, :])[None], torch.zeros_like(val), val)
        val = val.reshape(c, ph, gh, pw, gw)
        acc = input.new_zeros((c, ph, pw))
        for a in range(gh):                      # the operator accumulates iy-major, ix-minor, then divides
            for bb in range(gw):
                acc = acc + val[:, :, a, :, bb]
        out[r] = acc / count
    return out


# --------------------------------------------------------------------------------------------- detection/_utils.py
class BoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights, self.bbox_xform_clip = weights, bbox_xform_clip

    def decode(self, rel_codes, boxes):
        assert isinstance(boxes, (list, tuple))
        if isinstance(rel_codes, (list, tuple)):
            rel_codes = torch.cat(rel_codes, dim=0)
        boxes_per_image = [len(b) for b in boxes]
        concat_boxes = torch.cat(boxes, dim=0)
        pred = self.decode_single(rel_codes.reshape(sum(boxes_per_image), -1), concat_boxes)
        return pred.reshape(sum(boxes_per_image), -1, 4)

    def decode_single(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = rel_codes[:, 2::4] / ww
        dh = rel_codes[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.bbox_xform_clip)
        dh = torch.clamp(dh, max=self.bbox_xform_clip)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        pred_boxes = torch.zeros_like(rel_codes)
        pred_boxes[:, 0::4] = pred_ctr_x - 0.5 * pred_w
        pred_boxes[:, 1::4] = pred_ctr_y - 0.5 * pred_h
        pred_boxes[:, 2::4] = pred_ctr_x + 0.5 * pred_w
        pred_boxes[:, 3::4] = pred_ctr_y + 0.5 * pred_h
        return pred_boxes


# --------------------------------------------------------------------------------------------- detection/rpn.py
class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0)):
        super().__init__()
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        assert len(sizes) == len(aspect_ratios)
        self.sizes, self.aspect_ratios = sizes, aspect_ratios
        self.cell_anchors = None

    @staticmethod
    def generate_anchors(scales, aspect_ratios, dtype=torch.float32):
        scales = torch.as_tensor(scales, dtype=dtype)
        aspect_ratios = torch.as_tensor(aspect_ratios, dtype=dtype)
        h_ratios = torch.sqrt(aspect_ratios)
        w_ratios = 1 / h_ratios
        ws = (w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h_ratios[:, None] * scales[None, :]).view(-1)
        base_anchors = torch.stack([-ws, -hs, ws, hs], dim=1) / 2
        return base_anchors.round()

    def set_cell_anchors(self):
        if self.cell_anchors is None:
            self.cell_anchors = [self.generate_anchors(s, a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def grid_anchors(self, grid_sizes, strides):
        anchors = []
        for size, stride, base_anchors in zip(grid_sizes, strides, self.cell_anchors):
            grid_height, grid_width = size
            stride_height, stride_width = stride
            shifts_x = torch.arange(0, grid_width, dtype=torch.float32) * stride_width
            shifts_y = torch.arange(0, grid_height, dtype=torch.float32) * stride_height
            shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing='ij')
            shift_x, shift_y = shift_x.reshape(-1), shift_y.reshape(-1)
            shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
            anchors.append((shifts.view(-1, 1, 4) + base_anchors.view(1, -1, 4)).reshape(-1, 4))
        return anchors

    def forward(self, image_list, feature_maps):
        grid_sizes = tuple([fm.shape[-2:] for fm in feature_maps])
        image_size = image_list.tensors.shape[-2:]
        strides = tuple((image_size[0] / g[0], image_size[1] / g[1]) for g in grid_sizes)     # true quotients (0.4.2)
        self.set_cell_anchors()
        over_all = self.grid_anchors(grid_sizes, strides)
        return [torch.cat(list(over_all)) for _ in image_list.image_sizes]


class RPNHead(nn.Module):
    def __init__(self, in_channels, num_anchors):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for m in self.children():
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        logits, bbox_reg = [], []
        for feature in x:
            t = F.relu(self.conv(feature))
            logits.append(self.cls_logits(t))
            bbox_reg.append(self.bbox_pred(t))
        return logits, bbox_reg


def permute_and_flatten(layer, n, a, c, h, w):
    layer = layer.view(n, -1, c, h, w)
    layer = layer.permute(0, 3, 4, 1, 2)
    return layer.reshape(n, -1, c)


def concat_box_prediction_layers(box_cls, box_regression):
    box_cls_flattened, box_regression_flattened = [], []
    for box_cls_per_level, box_regression_per_level in zip(box_cls, box_regression):
        n, axc, h, w = box_cls_per_level.shape
        ax4 = box_regression_per_level.shape[1]
        a = ax4 // 4
        c = axc // a
        box_cls_flattened.append(permute_and_flatten(box_cls_per_level, n, a, c, h, w))
        box_regression_flattened.append(permute_and_flatten(box_regression_per_level, n, a, 4, h, w))
    box_cls = torch.cat(box_cls_flattened, dim=1).reshape(-1, c)
    box_regression = torch.cat(box_regression_flattened, dim=1).reshape(-1, 4)
    return box_cls, box_regression


class RegionProposalNetwork(nn.Module):
    def __init__(self, anchor_generator, head, fg_iou_thresh, bg_iou_thresh, batch_size_per_image,
                 positive_fraction, pre_nms_top_n, post_nms_top_n, nms_thresh):
        super().__init__()
        self.anchor_generator, self.head = anchor_generator, head
        self.box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self._pre_nms_top_n, self._post_nms_top_n, self.nms_thresh = pre_nms_top_n, post_nms_top_n, nms_thresh
        self.min_size = 1e-3

    @property
    def pre_nms_top_n(self):
        return self._pre_nms_top_n['training'] if self.training else self._pre_nms_top_n['testing']

    @property
    def post_nms_top_n(self):
        return self._post_nms_top_n['training'] if self.training else self._post_nms_top_n['testing']

    def _get_top_n_idx(self, objectness, num_anchors_per_level):
        r, offset = [], 0
        for ob in objectness.split(num_anchors_per_level, 1):
            num_anchors = ob.shape[1]
            pre_nms_top_n = min(self.pre_nms_top_n, num_anchors)
            _, top_n_idx = ob.topk(pre_nms_top_n, dim=1)
            r.append(top_n_idx + offset)
            offset += num_anchors
        return torch.cat(r, dim=1)

    def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
        num_images = proposals.shape[0]
        objectness = objectness.detach().reshape(num_images, -1)
        levels = torch.cat([torch.full((n,), idx, dtype=torch.int64) for idx, n in enumerate(num_anchors_per_level)], 0)
        levels = levels.reshape(1, -1).expand_as(objectness)
        top_n_idx = self._get_top_n_idx(objectness, num_anchors_per_level)
        batch_idx = torch.arange(num_images)[:, None]
        objectness = objectness[batch_idx, top_n_idx]
        levels = levels[batch_idx, top_n_idx]
        proposals = proposals[batch_idx, top_n_idx]
        final_boxes, final_scores = [], []
        for boxes, scores, lvl, img_shape in zip(proposals, objectness, levels, image_shapes):
            boxes = clip_boxes_to_image(boxes, img_shape)
            keep = remove_small_boxes(boxes, self.min_size)
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
            keep = batched_nms(boxes, scores, lvl, self.nms_thresh)
            keep = keep[:self.post_nms_top_n]
            final_boxes.append(boxes[keep])
            final_scores.append(scores[keep])
        return final_boxes, final_scores

    def forward(self, images, features, targets=None):
        if self.training:
            raise NotImplementedError('RPN training branch (losses) is not restated: every hnd/ghnd config distils '
                                      'with distill_backbone_only (src/models/org/rcnn.py:109-110)')
        features = list(features.values())
        objectness, pred_bbox_deltas = self.head(features)
        anchors = self.anchor_generator(images, features)
        num_images = len(anchors)
        num_anchors_per_level = [o[0].numel() for o in objectness]
        objectness, pred_bbox_deltas = concat_box_prediction_layers(objectness, pred_bbox_deltas)
        proposals = self.box_coder.decode(pred_bbox_deltas.detach(), anchors)
        proposals = proposals.view(num_images, -1, 4)
        boxes, scores = self.filter_proposals(proposals, objectness, images.image_sizes, num_anchors_per_level)
        self.last = {'objectness': objectness.reshape(num_images, -1), 'proposals': proposals, 'scores': scores,
                     'boxes': boxes}
        return boxes, {}


# --------------------------------------------------------------------------------------------- ops/poolers.py
class LevelMapper(object):
    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(torch.cat([box_area(b) for b in boxlists]))
        target_lvls = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        target_lvls = torch.clamp(target_lvls, min=self.k_min, max=self.k_max)
        return target_lvls.to(torch.int64) - self.k_min


class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names, output_size, sampling_ratio):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        self.featmap_names, self.sampling_ratio, self.output_size = featmap_names, sampling_ratio, tuple(output_size)
        self.scales, self.map_levels = None, None

    @staticmethod
    def convert_to_roi_format(boxes):
        concat_boxes = torch.cat(boxes, dim=0)
        ids = torch.cat([torch.full((len(b), 1), i, dtype=concat_boxes.dtype) for i, b in enumerate(boxes)], dim=0)
        return torch.cat([ids, concat_boxes], dim=1)

    @staticmethod
    def infer_scale(feature, original_size):
        size = feature.shape[-2:]
        possible_scales = []
        for s1, s2 in zip(size, original_size):
            approx_scale = float(s1) / s2
            possible_scales.append(2 ** torch.tensor(approx_scale).log2().round().item())
        assert possible_scales[0] == possible_scales[1]
        return possible_scales[0]

    def setup_scales(self, features, image_shapes):
        original_input_shape = tuple(max(s) for s in zip(*image_shapes))
        scales = [self.infer_scale(feat, original_input_shape) for feat in features]
        lvl_min, lvl_max = -math.log2(scales[0]), -math.log2(scales[-1])
        self.scales = scales
        self.map_levels = LevelMapper(lvl_min, lvl_max)

    def forward(self, x, boxes, image_shapes):
        x = [v for k, v in x.items() if k in self.featmap_names]
        num_levels = len(x)
        rois = self.convert_to_roi_format(boxes)
        if self.scales is None:
            self.setup_scales(x, image_shapes)
        if num_levels == 1:
            return roi_align(x[0], rois, self.output_size, self.scales[0], self.sampling_ratio)
        levels = self.map_levels(boxes)
        num_rois, num_channels = len(rois), x[0].shape[1]
        result = torch.zeros((num_rois, num_channels,) + self.output_size, dtype=x[0].dtype)
        for level, (per_level_feature, scale) in enumerate(zip(x, self.scales)):
            idx_in_level = torch.nonzero(levels == level).squeeze(1)
            result[idx_in_level] = roi_align(per_level_feature, rois[idx_in_level], self.output_size, scale,
                                             self.sampling_ratio)
        return result


# --------------------------------------------------------------------------------------------- faster_rcnn.py heads
class TwoMLPHead(nn.Module):
    def __init__(self, in_channels, representation_size):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)

    def forward(self, x):
        x = x.flatten(start_dim=1)
        x = F.relu(self.fc6(x))
        return F.relu(self.fc7(x))


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)

    def forward(self, x):
        if x.dim() == 4:
            assert list(x.shape[2:]) == [1, 1]
        x = x.flatten(start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)


# --------------------------------------------------------------------------------------------- roi_heads.py
def maskrcnn_inference(x, labels):
    """sigmoid of the mask logits, the channel of each detection's predicted class, split per image -> [n_i,1,M,M]"""
    mask_prob = x.sigmoid()
    num_masks = x.shape[0]
    boxes_per_image = [len(l) for l in labels]
    labels = torch.cat(labels)
    index = torch.arange(num_masks, device=labels.device)
    mask_prob = mask_prob[index, labels][:, None]
    return mask_prob.split(boxes_per_image, dim=0)


def expand_boxes(boxes, scale):
    w_half = (boxes[:, 2] - boxes[:, 0]) * .5
    h_half = (boxes[:, 3] - boxes[:, 1]) * .5
    x_c = (boxes[:, 2] + boxes[:, 0]) * .5
    y_c = (boxes[:, 3] + boxes[:, 1]) * .5
    w_half *= scale
    h_half *= scale
    boxes_exp = torch.zeros_like(boxes)
    boxes_exp[:, 0] = x_c - w_half
    boxes_exp[:, 2] = x_c + w_half
    boxes_exp[:, 1] = y_c - h_half
    boxes_exp[:, 3] = y_c + h_half
    return boxes_exp


def expand_masks(mask, padding):
    M = mask.shape[-1]
    scale = float(M + 2 * padding) / M
    padded_mask = F.pad(mask, (padding,) * 4)
    return padded_mask, scale


def paste_mask_in_image(mask, box, im_h, im_w):
    TO_REMOVE = 1
    w = int(box[2] - box[0] + TO_REMOVE)
    h = int(box[3] - box[1] + TO_REMOVE)
    w = max(w, 1)
    h = max(h, 1)
    mask = mask.expand((1, 1, -1, -1))                       # batch and channel dims
    mask = F.interpolate(mask, size=(h, w), mode='bilinear', align_corners=False)
    mask = mask[0][0]
    im_mask = torch.zeros((im_h, im_w), dtype=mask.dtype, device=mask.device)
    x_0 = max(box[0], 0)
    x_1 = min(box[2] + 1, im_w)
    y_0 = max(box[1], 0)
    y_1 = min(box[3] + 1, im_h)
    im_mask[y_0:y_1, x_0:x_1] = mask[(y_0 - box[1]):(y_1 - box[1]), (x_0 - box[0]):(x_1 - box[0])]
    return im_mask


def paste_masks_in_image(masks, boxes, img_shape, padding=1):
    masks, scale = expand_masks(masks, padding=padding)
    boxes = expand_boxes(boxes, scale).to(dtype=torch.int64).tolist()
    im_h, im_w = img_shape
    res = [paste_mask_in_image(m[0], b, im_h, im_w) for m, b in zip(masks, boxes)]
    if len(res) > 0:
        res = torch.stack(res, dim=0)[:, None]
    else:
        res = masks.new_empty((0, 1, im_h, im_w))
    return res


def heatmaps_to_keypoints(maps, rois):
    """(#rois, #keypoints, 3) with columns (x, y, 1) and the heatmap value at the argmax, per RoI"""
    offset_x = rois[:, 0]
    offset_y = rois[:, 1]
    widths = rois[:, 2] - rois[:, 0]
    heights = rois[:, 3] - rois[:, 1]
    widths = widths.clamp(min=1)
    heights = heights.clamp(min=1)
    widths_ceil = widths.ceil()
    heights_ceil = heights.ceil()
    num_keypoints = maps.shape[1]
    xy_preds = torch.zeros((len(rois), 3, num_keypoints), dtype=torch.float32, device=maps.device)
    end_scores = torch.zeros((len(rois), num_keypoints), dtype=torch.float32, device=maps.device)
    for i in range(len(rois)):
        roi_map_width = int(widths_ceil[i].item())
        roi_map_height = int(heights_ceil[i].item())
        width_correction = widths[i] / roi_map_width
        height_correction = heights[i] / roi_map_height
        roi_map = F.interpolate(maps[i][None], size=(roi_map_height, roi_map_width), mode='bicubic',
                                align_corners=False)[0]
        w = roi_map.shape[2]
        pos = roi_map.reshape(num_keypoints, -1).argmax(dim=1)
        x_int = pos % w
        y_int = (pos - x_int) // w                           # exact: pos - x_int is a multiple of w
        x = (x_int.float() + 0.5) * width_correction
        y = (y_int.float() + 0.5) * height_correction
        xy_preds[i, 0, :] = x + offset_x[i]
        xy_preds[i, 1, :] = y + offset_y[i]
        xy_preds[i, 2, :] = 1
        end_scores[i, :] = roi_map[torch.arange(num_keypoints), y_int, x_int]
    return xy_preds.permute(0, 2, 1), end_scores


def keypointrcnn_inference(x, boxes):
    kp_probs, kp_scores = [], []
    boxes_per_image = [len(box) for box in boxes]
    for xx, bb in zip(x.split(boxes_per_image, dim=0), boxes):
        kp_prob, scores = heatmaps_to_keypoints(xx, bb)
        kp_probs.append(kp_prob)
        kp_scores.append(scores)
    return kp_probs, kp_scores


class RoIHeads(nn.Module):
    def __init__(self, box_roi_pool, box_head, box_predictor, fg_iou_thresh, bg_iou_thresh,
                 batch_size_per_image, positive_fraction, bbox_reg_weights, score_thresh, nms_thresh,
                 detections_per_img, mask_roi_pool=None, mask_head=None, mask_predictor=None,
                 keypoint_roi_pool=None, keypoint_head=None, keypoint_predictor=None):
        super().__init__()
        if bbox_reg_weights is None:
            bbox_reg_weights = (10., 10., 5., 5.)
        self.box_coder = BoxCoder(bbox_reg_weights)
        self.box_roi_pool, self.box_head, self.box_predictor = box_roi_pool, box_head, box_predictor
        self.score_thresh, self.nms_thresh, self.detections_per_img = score_thresh, nms_thresh, detections_per_img
        self.mask_roi_pool, self.mask_head, self.mask_predictor = mask_roi_pool, mask_head, mask_predictor
        self.keypoint_roi_pool, self.keypoint_head = keypoint_roi_pool, keypoint_head
        self.keypoint_predictor = keypoint_predictor

    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
        num_classes = class_logits.shape[-1]
        boxes_per_image = [len(b) for b in proposals]
        pred_boxes = self.box_coder.decode(box_regression, proposals)
        pred_scores = F.softmax(class_logits, -1)
        pred_boxes = pred_boxes.split(boxes_per_image, 0)
        pred_scores = pred_scores.split(boxes_per_image, 0)
        all_boxes, all_scores, all_labels = [], [], []
        for boxes, scores, image_shape in zip(pred_boxes, pred_scores, image_shapes):
            boxes = clip_boxes_to_image(boxes, image_shape)
            labels = torch.arange(num_classes).view(1, -1).expand_as(scores)
            boxes, scores, labels = boxes[:, 1:], scores[:, 1:], labels[:, 1:]      # drop the background class
            boxes, scores, labels = boxes.reshape(-1, 4), scores.flatten(), labels.flatten()
            inds = torch.nonzero(scores > self.score_thresh).squeeze(1)
            boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
            keep = batched_nms(boxes, scores, labels, self.nms_thresh)
            keep = keep[:self.detections_per_img]
            all_boxes.append(boxes[keep])
            all_scores.append(scores[keep])
            all_labels.append(labels[keep])
        return all_boxes, all_scores, all_labels

    def forward(self, features, proposals, image_shapes, targets=None):
        if self.training:
            raise NotImplementedError('RoIHeads training branch (losses) is not restated (never run by hnd/ghnd)')
        box_features = self.box_roi_pool(features, proposals, image_shapes)
        box_features = self.box_head(box_features)
        class_logits, box_regression = self.box_predictor(box_features)
        self.last = {'class_logits': class_logits, 'box_regression': box_regression}
        self.last_image_shapes = list(image_shapes)
        boxes, scores, labels = self.postprocess_detections(class_logits, box_regression, proposals, image_shapes)
        result = [dict(boxes=boxes[i], labels=labels[i], scores=scores[i]) for i in range(len(boxes))]
        self.last_det_boxes = [b.clone() for b in boxes]         # in the resized image's frame (before postprocess)
        if self.mask_roi_pool is not None:                       # has_mask (eval branch)
            mask_proposals = [p['boxes'] for p in result]
            mask_features = self.mask_roi_pool(features, mask_proposals, image_shapes)
            mask_features = self.mask_head(mask_features)
            mask_logits = self.mask_predictor(mask_features)
            self.last['mask_logits'] = mask_logits
            masks_probs = maskrcnn_inference(mask_logits, [r['labels'] for r in result])
            for mask_prob, r in zip(masks_probs, result):
                r['masks'] = mask_prob
        if self.keypoint_roi_pool is not None:                   # has_keypoint (eval branch)
            keypoint_proposals = [p['boxes'] for p in result]
            keypoint_features = self.keypoint_roi_pool(features, keypoint_proposals, image_shapes)
            keypoint_features = self.keypoint_head(keypoint_features)
            keypoint_logits = self.keypoint_predictor(keypoint_features)
            self.last['keypoint_logits'] = keypoint_logits
            keypoints_probs, kp_scores = keypointrcnn_inference(keypoint_logits, keypoint_proposals)
            for keypoint_prob, kps, r in zip(keypoints_probs, kp_scores, result):
                r['keypoints'] = keypoint_prob
                r['keypoints_scores'] = kps
        return result, {}
