"""Eval-mode detector heads on the HIP path (the validation path).

What runs here is what ``src/models/org/rcnn.py:124-127`` reaches when ``distill_backbone_only`` is False:
``self.rpn(images, features)`` -> ``self.roi_heads(features, proposals, image_sizes)`` -> ``transform.postprocess``,
i.e. torchvision 0.4.2's RegionProposalNetwork / RoIHeads (box branch, and for Mask / Keypoint R-CNN the mask /
keypoint branches with ``paste_masks_in_image`` / ``heatmaps_to_keypoints``; restated for the CPU oracle in
oracle/tv042_det.py, which states the version-sensitive details this file follows: ascending-index ``nms`` result,
float anchor strides, no empty-box removal in ``postprocess_detections``, the Python-loop mask paste, bicubic keypoint
heat-map resize).

Arithmetic is libhnd_hip.so: the RPN head convs, fc6 (as a 7x7 valid conv over the pooled map), fc7 and the two
predictors on ``hnd_conv2d_igemm``, the branch convs on its Winograd form and the transposed convs on its data-gradient
form; anchors + box decoding, clipping, NMS (bit-exact kept set), RoIAlign, softmax as the kernels of csrc/detect.hip,
mask probabilities / paste / bilinear x2 / heat-maps-to-keypoints as those of csrc/detect_heads.hip; the variable-length
index bookkeeping -- per-level top-k, the score order of NMS, every ``nonzero`` compaction -- as those of
csrc/select.hip (stable descending argsort, ordered predicate compaction).  torch is used for storage, gathers by
those indices and concatenation, never for box or feature arithmetic beyond single exactly-rounded adds that are part
of the reference's own index trick (``batched_nms`` coordinate offsets).

Training-mode branches (RPN / RoI losses) do not exist: every hnd/ghnd config trains with ``org_loss_factor: 0``
behind ``distill_backbone_only`` (src/models/org/rcnn.py:109-110), so they never run in the reference either.
"""
import ctypes as C
import math

import torch
from torch import nn

from . import _lib, engine as E, ops

_L = _lib.load()
XFORM_CLIP = math.log(1000.0 / 16)


def _check(rc, what):
    if rc:
        _lib.check(rc, what)


# --------------------------------------------------------------------------------------------- box ops (index work)
def clip_boxes_(boxes, size):
    """in place ops.boxes.clip_boxes_to_image on a contiguous [n, 4] device tensor"""
    _check(_L.hnd_clip_boxes(boxes.data_ptr(), boxes.shape[0], float(size[0]), float(size[1]), ops.stream_ptr()),
           'hnd_clip_boxes')
    return boxes


def _compacted(call, n, device):
    """run one hnd_nonzero_* launch and return the indices it kept (ascending), like torch.nonzero(p).squeeze(1);
    the count comes back through the same kind of host read torch.nonzero needs for its output size"""
    out = torch.empty(max(int(n), 1), dtype=torch.int64, device=device)
    count = torch.empty(1, dtype=torch.int64, device=device)
    call(out.data_ptr(), count.data_ptr())
    return out[:int(count.item())]


def nonzero_flags(flags):
    f = flags.contiguous()
    return _compacted(lambda o, c: _check(_L.hnd_nonzero_u8(f.data_ptr(), f.numel(), o, c, ops.stream_ptr()),
                                          'hnd_nonzero_u8'), f.numel(), f.device)


def nonzero_greater(x, threshold):
    x = x.contiguous()
    return _compacted(lambda o, c: _check(_L.hnd_nonzero_gt_f32(x.data_ptr(), x.numel(), float(threshold), o, c,
                                                                ops.stream_ptr()), 'hnd_nonzero_gt_f32'),
                      x.numel(), x.device)


def nonzero_equal(x, value):
    x = x.contiguous()
    return _compacted(lambda o, c: _check(_L.hnd_nonzero_eq_i64(x.data_ptr(), x.numel(), int(value), o, c,
                                                                ops.stream_ptr()), 'hnd_nonzero_eq_i64'),
                      x.numel(), x.device)


def argsort_descending(keys):
    """stable descending argsort of a 1-D fp32 tensor (== torch.sort(descending=True, stable=True)[1])"""
    keys = keys.contiguous()
    n = keys.numel()
    order = torch.empty(n, dtype=torch.int64, device=keys.device)
    if n:
        ws = torch.empty(int(_L.hnd_argsort_desc_workspace(n)), dtype=torch.uint8, device=keys.device)
        _check(_L.hnd_argsort_desc_f32(keys.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ops.stream_ptr()),
               'hnd_argsort_desc_f32')
    return order


def remove_small_boxes(boxes, min_size):
    b = boxes.contiguous()
    return _compacted(lambda o, c: _check(_L.hnd_nonzero_min_size(b.data_ptr(), b.shape[0], float(min_size), o, c,
                                                                  ops.stream_ptr()), 'hnd_nonzero_min_size'),
                      b.shape[0], b.device)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms with 0.4.2's CPU-operator result order: kept indices ASCENDING (oracle/tv042_det.py)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.contiguous()
    order = argsort_descending(scores)
    ws = torch.empty(int(_L.hnd_nms_workspace(n)), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty(n, dtype=torch.uint8, device=boxes.device)
    _check(_L.hnd_nms(boxes.data_ptr(), order.data_ptr(), n, float(iou_threshold), ws.data_ptr(), keep.data_ptr(),
                      ops.stream_ptr()), 'hnd_nms')
    return nonzero_flags(keep)


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)          # the reference's per-group coordinate offset trick
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def resize_boxes(boxes, original_size, new_size):
    rh, rw = (float(s) / float(o) for s, o in zip(new_size, original_size))
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * rw, ymin * rh, xmax * rw, ymax * rh), dim=1)


# --------------------------------------------------------------------------------------------- RPN
class AnchorGenerator(nn.Module):
    """rpn.py AnchorGenerator: only the cell anchors live on the host; the grid is generated inside hnd_rpn_decode."""

    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0)):
        super().__init__()
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        self.sizes, self.aspect_ratios = sizes, aspect_ratios

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def cell_anchors(self):
        """generate_anchors (host, a dozen numbers per level): fp32 arithmetic in the reference's order"""
        out = []
        for scales, ratios in zip(self.sizes, self.aspect_ratios):
            scales = torch.as_tensor(scales, dtype=torch.float32)
            ratios = torch.as_tensor(ratios, dtype=torch.float32)
            h_ratios = torch.sqrt(ratios)
            w_ratios = 1 / h_ratios
            ws = (w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h_ratios[:, None] * scales[None, :]).view(-1)
            out.append((torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round())
        return out


class RPNHead(nn.Module):
    """parameter holder with torchvision's names; executed by RpnEngine (3x3 conv + ReLU, then cls_logits and
    bbox_pred as ONE 1x1 conv whose output channels are [A logits | 4A deltas])."""

    def __init__(self, in_channels, num_anchors):
        super().__init__()
        from .hipnn import Conv2d
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls_logits = Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for m in self.children():
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)


class RpnEngine(object):
    def __init__(self, head):
        self.head = head
        self.wc3 = E.WeightCache(head.conv.weight)
        tile = E.use_winograd(head.conv.weight.shape[1], head.conv.weight.shape[0], 1)
        self.wino = E.WinoCache(head.conv.weight, tile) if tile else None
        self.bufs = None
        self.plan_key = None
        self.merged = None

    def _merged_1x1(self):
        """[cls_logits; bbox_pred] as one OIHW weight / bias (re-made when either parameter changes)"""
        h = self.head
        ver = tuple(E.weight_version(p) for p in (h.cls_logits.weight, h.cls_logits.bias, h.bbox_pred.weight,
                                                  h.bbox_pred.bias))
        if self.merged is None or self.merged[0] != ver:
            w = torch.cat([h.cls_logits.weight.detach(), h.bbox_pred.weight.detach()], 0).contiguous()
            b = torch.cat([h.cls_logits.bias.detach(), h.bbox_pred.bias.detach()], 0).contiguous()
            if self.merged is None:
                self.merged = [ver, w, b, ops.pack_weights(w)]
            else:
                self.merged[1].copy_(w)
                self.merged[2].copy_(b)
                self.merged[3].repack()
                self.merged[0] = ver
        return self.merged

    def forward(self, feats):
        """feats: NHWC pyramid maps.  Returns the per-level [N, H, W, ldc] head outputs (logits | deltas)."""
        if self.bufs is None:
            self.bufs = E.Buffers(feats[0].device)
        _, w1, b1, pk1 = self._merged_1x1()
        if self.wino is not None:
            self.wino.get(False)
            self.wino.refresh()
        else:
            self.wc3.get()
            self.wc3.refresh()
        key = tuple((f.data_ptr(), tuple(f.shape)) for f in feats) + (self.head.conv.bias.data_ptr(),)
        if key != self.plan_key:
            self.plan, self.outs = [], []
            cout = w1.shape[0]
            ldc = ops.round_up(cout, 4)
            need = (0, 0)
            for i, f in enumerate(feats):
                n, h, w, c = f.shape
                t = self.bufs.get('t%d' % i, (n, h, w, c))
                o = self.bufs.get('o%d' % i, (n, h, w, ldc))
                bias3 = self.head.conv.bias.detach()
                if self.wino is not None:
                    nv, nm = ops.WinoConv.scratch_elems(n, h, w, c, c, E.WINOGRAD)
                    need = (max(need[0], nv), max(need[1], nm))
                    v, m = self.bufs.get('wino_v', (need[0],)), self.bufs.get('wino_m', (need[1],))
                    self.plan += ops.WinoConv(f, self.wino.get(False), t, v, m, epi_shift=bias3,
                                              relu=True).launches('rpn.conv%d' % i)
                else:
                    self.plan.append((ops.conv_forward(f, self.wc3.get(), t, 3, 1, 1, epi_shift=bias3, relu=True),
                                      'rpn.conv%d' % i))
                self.plan.append((ops.conv_forward(t, pk1, o, 1, 1, 0, epi_shift=b1, cout=cout), 'rpn.pred%d' % i))
                self.outs.append(o)
            self.plan_key = key
        for l, tag in self.plan:
            E._run(l, tag)
        return self.outs


class RegionProposalNetwork(nn.Module):
    def __init__(self, anchor_generator, head, fg_iou_thresh=0.7, bg_iou_thresh=0.3, batch_size_per_image=256,
                 positive_fraction=0.5, pre_nms_top_n=None, post_nms_top_n=None, nms_thresh=0.7):
        super().__init__()
        self.anchor_generator, self.head = anchor_generator, head
        self._pre_nms_top_n = pre_nms_top_n or dict(training=2000, testing=1000)
        self._post_nms_top_n = post_nms_top_n or dict(training=2000, testing=1000)
        self.nms_thresh, self.min_size = nms_thresh, 1e-3
        self._engine = None
        self.last = None

    @property
    def pre_nms_top_n(self):
        return self._pre_nms_top_n['training'] if self.training else self._pre_nms_top_n['testing']

    @property
    def post_nms_top_n(self):
        return self._post_nms_top_n['training'] if self.training else self._post_nms_top_n['testing']

    def decode(self, images, feats):
        """RPN head + anchors + BoxCoder.decode for the whole batch: objectness [N, total], proposals [N, total, 4]"""
        if self._engine is None:
            self._engine = RpnEngine(self.head)
        outs = self._engine.forward(feats)
        n = feats[0].shape[0]
        a = self.anchor_generator.num_anchors_per_location()[0]
        per_level = [f.shape[1] * f.shape[2] * a for f in feats]
        total = sum(per_level)
        dev = feats[0].device
        bufs = self._engine.bufs
        objectness, proposals = bufs.get('objectness', (n, total)), bufs.get('proposals', (n, total, 4))
        img_h, img_w = images.tensors.shape[-2:]
        offset = 0
        for f, o, base, cnt in zip(feats, outs, self.anchor_generator.cell_anchors(), per_level):
            h, w = f.shape[1], f.shape[2]
            stride_h, stride_w = img_h / h, img_w / w               # rpn.py: true quotients (floats)
            flat = (C.c_float * (a * 4))(*[float(v) for v in base.reshape(-1)])
            _check(_L.hnd_rpn_decode(o.data_ptr(), n, h, w, o.shape[3], a, flat, stride_h, stride_w, offset, total,
                                     XFORM_CLIP, objectness.data_ptr(), proposals.data_ptr(), ops.stream_ptr()),
                   'hnd_rpn_decode')
            offset += cnt
        return objectness, proposals, per_level

    def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
        num_images = proposals.shape[0]
        dev = proposals.device
        levels = torch.cat([torch.full((n,), idx, dtype=torch.int64, device=dev)
                            for idx, n in enumerate(num_anchors_per_level)], 0)
        levels = levels.reshape(1, -1).expand_as(objectness)
        r, offset = [], 0
        for ob in objectness.split(num_anchors_per_level, 1):       # top-k per level, independently
            k = min(self.pre_nms_top_n, ob.shape[1])
            # top-k = the head of the stable descending order (hnd_argsort_desc_f32), image by image
            r.append(torch.stack([argsort_descending(row)[:k] for row in ob]) + offset)
            offset += ob.shape[1]
        top_n_idx = torch.cat(r, dim=1)
        batch_idx = torch.arange(num_images, device=dev)[:, None]
        objectness, levels = objectness[batch_idx, top_n_idx], levels[batch_idx, top_n_idx]
        proposals = proposals[batch_idx, top_n_idx]
        final_boxes, final_scores = [], []
        for boxes, scores, lvl, img_shape in zip(proposals, objectness, levels, image_shapes):
            boxes = clip_boxes_(boxes.contiguous(), img_shape)
            keep = remove_small_boxes(boxes, self.min_size)
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
            keep = batched_nms(boxes, scores, lvl, self.nms_thresh)
            keep = keep[:self.post_nms_top_n]
            final_boxes.append(boxes[keep])
            final_scores.append(scores[keep])
        return final_boxes, final_scores

    def forward(self, images, features, targets=None):
        if self.training:
            raise NotImplementedError('RPN training branch (proposal losses): never run by the hnd/ghnd configs '
                                      '(org_loss_factor 0 behind distill_backbone_only, rcnn.py:109-110)')
        from .hipnn import to_nhwc
        feats = [to_nhwc(v) for v in features.values()]
        objectness, proposals, per_level = self.decode(images, feats)
        boxes, scores = self.filter_proposals(proposals, objectness, images.image_sizes, per_level)
        self.last = {'objectness': objectness, 'proposals': proposals, 'scores': scores}
        return boxes, {}


# --------------------------------------------------------------------------------------------- RoI heads
class MultiScaleRoIAlign(nn.Module):
    """ops/poolers.py: FPN level by box scale (LevelMapper), then hnd_roi_align per level into one NHWC result"""

    def __init__(self, featmap_names, output_size, sampling_ratio):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        self.featmap_names, self.output_size, self.sampling_ratio = featmap_names, tuple(output_size), sampling_ratio
        self.scales, self.k_min, self.k_max = None, None, None

    def setup_scales(self, feats, image_shapes):
        original = tuple(max(s) for s in zip(*image_shapes))
        scales = []
        for f in feats:
            poss = [2 ** torch.tensor(float(s1) / s2).log2().round().item() for s1, s2 in zip(f.shape[1:3], original)]
            assert poss[0] == poss[1]
            scales.append(poss[0])
        self.scales = scales
        self.k_min, self.k_max = -math.log2(scales[0]), -math.log2(scales[-1])

    def map_levels(self, rois):
        area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
        s = torch.sqrt(area)
        lv = torch.floor(4 + torch.log2(s / 224 + 1e-6))
        return torch.clamp(lv, min=self.k_min, max=self.k_max).to(torch.int64) - int(self.k_min)

    def forward(self, x, boxes, image_shapes):
        """x: OrderedDict of logical NCHW maps; returns the pooled NHWC buffer [K, ph, pw, C] and the rois [K, 5]"""
        from .hipnn import to_nhwc
        feats = [to_nhwc(v) for k, v in x.items() if k in self.featmap_names]
        dev = feats[0].device
        ids = torch.cat([torch.full((len(b), 1), float(i), dtype=torch.float32, device=dev)
                         for i, b in enumerate(boxes)], 0)
        rois = torch.cat([ids, torch.cat(boxes, 0)], 1).contiguous()
        if self.scales is None:
            self.setup_scales(feats, image_shapes)
        k, c = rois.shape[0], feats[0].shape[3]
        ph, pw = self.output_size
        out = torch.zeros((k, ph, pw, c), dtype=torch.float32, device=dev)
        levels = self.map_levels(rois) if len(feats) > 1 else torch.zeros(k, dtype=torch.int64, device=dev)
        for level, (f, scale) in enumerate(zip(feats, self.scales)):
            idx = nonzero_equal(levels, level)
            _check(_L.hnd_roi_align(f.data_ptr(), f.shape[0], f.shape[1], f.shape[2], c, rois.data_ptr(),
                                    idx.data_ptr(), idx.numel(), float(scale), ph, pw, self.sampling_ratio,
                                    out.data_ptr(), ops.stream_ptr()), 'hnd_roi_align')
        return out, rois


class TwoMLPHead(nn.Module):
    def __init__(self, in_channels, representation_size):
        super().__init__()
        from .hipnn import Linear
        self.fc6 = Linear(in_channels, representation_size)
        self.fc7 = Linear(representation_size, representation_size)


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        from .hipnn import Linear
        self.cls_score = Linear(in_channels, num_classes)
        self.bbox_pred = Linear(in_channels, num_classes * 4)


class _LinearAsConv(object):
    """nn.Linear over a pooled NHWC map: weight [out, C*kh*kw] in NCHW-flatten order == an OIHW conv weight"""

    def __init__(self, linear, c, kh, kw):
        self.linear, self.shape = linear, (linear.weight.shape[0], c, kh, kw)
        self.view = linear.weight.detach().view(self.shape)
        self.wc = E.WeightCache(self.view)
        self.ptr = linear.weight.data_ptr()

    def pack(self):
        w = self.linear.weight
        if w.data_ptr() != self.ptr:                        # parameter storage moved (e.g. load_state_dict re-point)
            self.view = w.detach().view(self.shape)
            self.wc = E.WeightCache(self.view)
            self.ptr = w.data_ptr()
        pk = self.wc.get()
        ver = E.weight_version(w)
        if getattr(self, 'ver', None) != ver:
            self.wc.refresh(force=True)
            self.ver = ver
        return pk


class RoIHeads(nn.Module):
    def __init__(self, box_roi_pool, box_head, box_predictor, fg_iou_thresh=0.5, bg_iou_thresh=0.5,
                 batch_size_per_image=512, positive_fraction=0.25, bbox_reg_weights=None, score_thresh=0.05,
                 nms_thresh=0.5, detections_per_img=100, mask_roi_pool=None, mask_head=None, mask_predictor=None,
                 keypoint_roi_pool=None, keypoint_head=None, keypoint_predictor=None):
        super().__init__()
        self.box_roi_pool, self.box_head, self.box_predictor = box_roi_pool, box_head, box_predictor
        self.bbox_reg_weights = bbox_reg_weights or (10., 10., 5., 5.)
        self.score_thresh, self.nms_thresh, self.detections_per_img = score_thresh, nms_thresh, detections_per_img
        self.mask_roi_pool, self.mask_head, self.mask_predictor = mask_roi_pool, mask_head, mask_predictor
        self.keypoint_roi_pool, self.keypoint_head = keypoint_roi_pool, keypoint_head
        self.keypoint_predictor = keypoint_predictor
        self._lin = None
        self._wc = None
        self.last = None

    def box_branch(self, pooled):
        """TwoMLPHead + FastRCNNPredictor on the pooled NHWC map [K, 7, 7, C] -> class_logits [K, ncls],
        box_regression [K, 4 ncls]"""
        k, ph, pw, c = pooled.shape
        if self._lin is None:
            self._lin = (_LinearAsConv(self.box_head.fc6, c, ph, pw),
                         _LinearAsConv(self.box_head.fc7, self.box_head.fc6.weight.shape[0], 1, 1),
                         _LinearAsConv(self.box_predictor.cls_score, self.box_head.fc7.weight.shape[0], 1, 1),
                         _LinearAsConv(self.box_predictor.bbox_pred, self.box_head.fc7.weight.shape[0], 1, 1))
        fc6, fc7, cls, reg = self._lin
        dev = pooled.device
        rep = fc6.shape[0]
        ncls = cls.shape[0]
        h6 = torch.empty((k, 1, 1, rep), dtype=torch.float32, device=dev)
        h7 = torch.empty((k, 1, 1, fc7.shape[0]), dtype=torch.float32, device=dev)
        logits = torch.empty((k, 1, 1, ncls), dtype=torch.float32, device=dev)
        deltas = torch.empty((k, 1, 1, reg.shape[0]), dtype=torch.float32, device=dev)
        ops.conv_forward(pooled, fc6.pack(), h6, (ph, pw), 1, 0, epi_shift=self.box_head.fc6.bias.detach(),
                         relu=True).run()
        ops.conv_forward(h6, fc7.pack(), h7, 1, 1, 0, epi_shift=self.box_head.fc7.bias.detach(), relu=True).run()
        ops.conv_forward(h7, cls.pack(), logits, 1, 1, 0, epi_shift=self.box_predictor.cls_score.bias.detach()).run()
        ops.conv_forward(h7, reg.pack(), deltas, 1, 1, 0, epi_shift=self.box_predictor.bbox_pred.bias.detach()).run()
        return logits.view(k, ncls), deltas.view(k, reg.shape[0])

    def postprocess_detections(self, class_logits, box_regression, rois, boxes_per_image, image_shapes):
        dev = class_logits.device
        k, ncls = class_logits.shape
        scores_all = torch.empty_like(class_logits)
        ops.softmax_rows(class_logits, scores_all)
        hw = torch.tensor([[float(s[0]), float(s[1])] for s in image_shapes], dtype=torch.float32, device=dev)
        boxes_all = torch.empty((k, ncls, 4), dtype=torch.float32, device=dev)
        wx, wy, ww, wh = self.bbox_reg_weights
        _check(_L.hnd_box_decode_clip(box_regression.data_ptr(), box_regression.shape[1], rois.data_ptr(),
                                      hw.data_ptr(), k, ncls, wx, wy, ww, wh, XFORM_CLIP, boxes_all.data_ptr(),
                                      ops.stream_ptr()), 'hnd_box_decode_clip')
        all_boxes, all_scores, all_labels = [], [], []
        for boxes, scores in zip(boxes_all.split(boxes_per_image, 0), scores_all.split(boxes_per_image, 0)):
            labels = torch.arange(ncls, device=dev).view(1, -1).expand_as(scores)
            boxes, scores, labels = boxes[:, 1:], scores[:, 1:], labels[:, 1:]          # drop background
            boxes, scores, labels = boxes.reshape(-1, 4), scores.flatten(), labels.flatten()
            inds = nonzero_greater(scores, self.score_thresh)
            boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
            keep = batched_nms(boxes, scores, labels, self.nms_thresh)
            keep = keep[:self.detections_per_img]
            all_boxes.append(boxes[keep])
            all_scores.append(scores[keep])
            all_labels.append(labels[keep])
        return all_boxes, all_scores, all_labels

    def forward(self, features, proposals, image_shapes, targets=None):
        if self.training:
            raise NotImplementedError('RoIHeads training branch (detection losses): never run by the hnd/ghnd configs')
        if sum(len(p) for p in proposals) == 0:         # nothing proposed (e.g. every box degenerate): empty detections
            dev = proposals[0].device
            empty = dict(boxes=torch.empty(0, 4, device=dev), labels=torch.empty(0, dtype=torch.int64, device=dev),
                         scores=torch.empty(0, device=dev))
            if self.mask_roi_pool is not None:
                m = 2 * self.mask_roi_pool.output_size[0]
                empty['masks'] = torch.empty(0, 1, m, m, device=dev)
            if self.keypoint_roi_pool is not None:
                nkp = self.keypoint_predictor.out_channels
                empty['keypoints'] = torch.empty(0, nkp, 3, device=dev)
                empty['keypoints_scores'] = torch.empty(0, nkp, device=dev)
            return [dict(empty) for _ in proposals], {}
        pooled, rois = self.box_roi_pool(features, proposals, image_shapes)
        class_logits, box_regression = self.box_branch(pooled)
        self.last = {'class_logits': class_logits, 'box_regression': box_regression, 'pooled': pooled}
        boxes, scores, labels = self.postprocess_detections(class_logits, box_regression, rois,
                                                            [len(p) for p in proposals], image_shapes)
        result = [dict(boxes=boxes[i], labels=labels[i], scores=scores[i]) for i in range(len(boxes))]
        if self.mask_roi_pool is not None:                       # roi_heads.py has_mask, eval branch
            probs = self.mask_branch(features, boxes, labels, image_shapes)
            for r, p in zip(result, probs):
                r['masks'] = p
        if self.keypoint_roi_pool is not None:                   # roi_heads.py has_keypoint, eval branch
            kps, kp_scores = self.keypoint_branch(features, boxes, image_shapes)
            for r, kp, sc in zip(result, kps, kp_scores):
                r['keypoints'] = kp
                r['keypoints_scores'] = sc
        return result, {}

    # ------------------------------------------------------------------ mask / keypoint branches (eval)
    def _cache(self, key, weight):
        if self._wc is None:
            self._wc = {}
        wc = self._wc.get(key)
        if wc is None or wc.weight is not weight:
            wc = self._wc[key] = E.WeightCache(weight)
        return wc

    def _conv3x3_relu_chain(self, x, convs, tag):
        """[Conv2d(3x3, pad 1) + bias + ReLU] * len(convs) on an NHWC buffer: Winograd F(4x4,3x3) (input transform ->
        36 GEMMs in one hnd_conv2d_igemm launch -> output transform with bias + ReLU) where engine.use_winograd says it
        pays (>= 128 channels: every branch conv of torchvision's heads), one direct launch otherwise"""
        for i, conv in enumerate(convs):
            assert conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) \
                and conv.dilation == (1, 1), 'the branch convs of torchvision 0.4.2 are 3x3 / stride 1 / pad 1'
            k, h, w, cin = x.shape
            cout = conv.weight.shape[0]
            y = torch.empty((k, h, w, cout), dtype=torch.float32, device=x.device)
            tile = E.use_winograd(cin, cout, 1)
            if tile:
                if self._wc is None:
                    self._wc = {}
                wn = self._wc.get((tag, i, 'wino'))
                if wn is None or wn.weight is not conv.weight:
                    wn = self._wc[(tag, i, 'wino')] = E.WinoCache(conv.weight, tile)
                ww = wn.get(False)
                wn.refresh()
                nv, nm = ops.WinoConv.scratch_elems(k, h, w, cin, cout, tile)
                v = torch.empty(nv, dtype=torch.float32, device=x.device)
                m = torch.empty(nm, dtype=torch.float32, device=x.device)
                for l, t in ops.WinoConv(x, ww, y, v, m, epi_shift=conv.bias.detach(),
                                         relu=True).launches('%s.conv%d' % (tag, i)):
                    E._run(l, t)
            else:
                wc = self._cache((tag, i), conv.weight)
                pk = wc.get()
                wc.refresh()
                E._run(ops.conv_forward(x, pk, y, 3, 1, 1, epi_shift=conv.bias.detach(), relu=True),
                       '%s.conv%d' % (tag, i))
            x = y
        return x

    def _conv_transpose(self, x, convt, tag, relu):
        """nn.ConvTranspose2d == the data gradient of the conv with the same weight tensor: one dense launch per
        output parity (ops.conv_dgrad), bias (+ ReLU) in the epilogue"""
        k, h, w, _ = x.shape
        ks, st, pd = convt.kernel_size[0], convt.stride[0], convt.padding[0]
        oh, ow = (h - 1) * st - 2 * pd + ks, (w - 1) * st - 2 * pd + ks
        y = torch.empty((k, oh, ow, convt.weight.shape[1]), dtype=torch.float32, device=x.device)
        wc = self._cache((tag, 'T'), convt.weight)
        launches, _ = ops.conv_dgrad(x, wc, y, ks, st, pd, epi_shift=convt.bias.detach(), relu=relu)
        wc.refresh()
        for l in launches:
            E._run(l, tag)
        return y

    def mask_logits(self, features, det_boxes, image_shapes):
        """mask_roi_pool -> mask_head (4 x conv3x3 + ReLU) -> conv5_mask (2x2 stride-2 transposed conv + ReLU) ->
        mask_fcn_logits (1x1): NHWC [K, 28, 28, num_classes]"""
        pooled, _ = self.mask_roi_pool(features, det_boxes, image_shapes)
        convs = [m for m in self.mask_head if hasattr(m, 'weight')]
        x = self._conv3x3_relu_chain(pooled, convs, 'mask_head')
        x = self._conv_transpose(x, self.mask_predictor.conv5_mask, 'mask.conv5', relu=True)
        fin = self.mask_predictor.mask_fcn_logits
        wc = self._cache(('mask.logits', 0), fin.weight)
        pk = wc.get()
        wc.refresh()
        k, h, w, _ = x.shape
        logits = torch.empty((k, h, w, fin.weight.shape[0]), dtype=torch.float32, device=x.device)
        E._run(ops.conv_forward(x, pk, logits, 1, 1, 0, epi_shift=fin.bias.detach()), 'mask.logits')
        return logits

    def mask_branch(self, features, det_boxes, labels, image_shapes):
        """maskrcnn_inference: per image [n_i, 1, M, M] probabilities of each detection's predicted class"""
        per_image = [len(b) for b in det_boxes]
        dev = det_boxes[0].device
        if sum(per_image) == 0:
            m = 2 * self.mask_roi_pool.output_size[0]
            return [torch.empty((0, 1, m, m), dtype=torch.float32, device=dev) for _ in det_boxes]
        logits = self.mask_logits(features, det_boxes, image_shapes)
        k, m, _, ldc = logits.shape
        lab = torch.cat(labels).to(torch.int64).contiguous()
        probs = torch.empty((k, 1, m, m), dtype=torch.float32, device=dev)
        _check(_L.hnd_mask_probs(logits.data_ptr(), lab.data_ptr(), k, m, ldc, probs.data_ptr(), ops.stream_ptr()),
               'hnd_mask_probs')
        if self.last is not None:
            self.last['mask_logits'] = logits
        return list(probs.split(per_image, 0))

    def keypoint_logits(self, features, det_boxes, image_shapes):
        """keypoint_roi_pool -> keypoint_head (8 x conv3x3 + ReLU) -> kps_score_lowres (4x4 stride-2 transposed conv)
        -> bilinear x2: NHWC [K, 56, 56, num_keypoints]"""
        pooled, _ = self.keypoint_roi_pool(features, det_boxes, image_shapes)
        convs = [m for m in self.keypoint_head if hasattr(m, 'weight')]
        x = self._conv3x3_relu_chain(pooled, convs, 'keypoint_head')
        pred = self.keypoint_predictor
        low = self._conv_transpose(x, pred.kps_score_lowres, 'keypoint.lowres', relu=False)
        k, h, w, c = low.shape
        f = int(pred.up_scale)
        up = torch.empty((k, h * f, w * f, c), dtype=torch.float32, device=low.device)
        _check(_L.hnd_upsample_bilinear_nhwc(low.data_ptr(), k, h, w, c, f, up.data_ptr(), ops.stream_ptr()),
               'hnd_upsample_bilinear_nhwc')
        return up

    def keypoint_branch(self, features, det_boxes, image_shapes):
        """keypointrcnn_inference: per image keypoints [n_i, num_keypoints, 3] = (x, y, 1) and their heatmap scores"""
        per_image = [len(b) for b in det_boxes]
        dev = det_boxes[0].device
        nkp = self.keypoint_predictor.out_channels
        if sum(per_image) == 0:
            return ([torch.empty((0, nkp, 3), dtype=torch.float32, device=dev) for _ in det_boxes],
                    [torch.empty((0, nkp), dtype=torch.float32, device=dev) for _ in det_boxes])
        maps = self.keypoint_logits(features, det_boxes, image_shapes)
        k, h, w, ldc = maps.shape
        rois = torch.cat(det_boxes, 0).to(torch.float32).contiguous()
        xy = torch.empty((k, nkp, 3), dtype=torch.float32, device=dev)
        sc = torch.empty((k, nkp), dtype=torch.float32, device=dev)
        _check(_L.hnd_heatmaps_to_keypoints(maps.data_ptr(), k, h, w, ldc, nkp, rois.data_ptr(), xy.data_ptr(),
                                            sc.data_ptr(), ops.stream_ptr()), 'hnd_heatmaps_to_keypoints')
        if self.last is not None:
            self.last['keypoint_logits'] = maps
        return list(xy.split(per_image, 0)), list(sc.split(per_image, 0))


def paste_masks_in_image(masks, boxes, img_shape, padding=1):
    """transform.py postprocess -> roi_heads.paste_masks_in_image: masks [n, 1, M, M] probabilities, boxes [n, 4] in
    the original image's frame -> [n, 1, im_h, im_w].  expand_boxes / the int64 truncation are the reference's tensor
    expressions (fp32 element-wise ops: identical on any IEEE device); padding, bilinear resize and paste are one
    hnd_paste_masks launch."""
    assert padding == 1
    n, m = masks.shape[0], masks.shape[-1]
    im_h, im_w = int(img_shape[0]), int(img_shape[1])
    out = torch.empty((n, 1, im_h, im_w), dtype=torch.float32, device=masks.device)
    if n == 0:
        return out
    scale = float(m + 2 * padding) / m
    w_half = (boxes[:, 2] - boxes[:, 0]) * .5
    h_half = (boxes[:, 3] - boxes[:, 1]) * .5
    x_c = (boxes[:, 2] + boxes[:, 0]) * .5
    y_c = (boxes[:, 3] + boxes[:, 1]) * .5
    w_half = w_half * scale
    h_half = h_half * scale
    exp = torch.stack([x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half], 1).to(torch.int64).contiguous()
    probs = masks.contiguous()
    _check(_L.hnd_paste_masks(probs.data_ptr(), exp.data_ptr(), n, m, im_h, im_w, out.data_ptr(), ops.stream_ptr()),
           'hnd_paste_masks')
    return out


def resize_keypoints(keypoints, original_size, new_size):
    """transform.py resize_keypoints: x by new_w / orig_w, y by new_h / orig_h (python-float ratios)"""
    ratio_h, ratio_w = (float(s) / float(o) for s, o in zip(new_size, original_size))
    out = keypoints.clone()
    out[..., 0] *= ratio_w
    out[..., 1] *= ratio_h
    return out
