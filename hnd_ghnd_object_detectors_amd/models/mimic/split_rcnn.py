"""Head / tail split of a distilled detector (mirror of the reference's src/models/mimic/split_rcnn.py).

``RcnnHead`` is what runs next to the camera: transform -> stem -> layer1 encoder [-> neural filter gate]
[-> Quantizer]; its output ``z`` (3 channels at b3ch, uint8 when quantised) is what crosses the link.
``RcnnTail`` continues on the server: [Dequantizer ->] layer1 decoder -> layer2..4 -> FPN -> RPN -> RoI box head ->
post-processing (reference :186-196; the eval-mode detector of detection.py); with ``features_only`` set it stops at
the pyramid.
Both halves run the same HIP engines as the unsplit model, in eval mode (inference deployment).
"""
from collections import OrderedDict

from torch import nn

from ... import engine as E
from ...hipnn import ImageList, attach, to_nhwc
from ...structure.transformer import Compose, Dequantizer, Quantizer


class RcnnHead(nn.Module):
    def __init__(self, rcnn_model, bottleneck_transformer=None):
        super().__init__()
        body = rcnn_model.backbone.body
        self.transform = rcnn_model.transform
        self.layer0 = nn.Sequential(body['conv1'], body['bn1'], body['relu'], body['maxpool'])
        self.layer1_encoder = body['layer1'].encoder
        self.bottleneck_transformer = bottleneck_transformer
        self._stem = E.StemEngine(body['conv1'], body['bn1'])
        self._head = body['layer1'].head_engine()           # shared with the tail: same weights, separate plans

    def forward(self, images, targets=None):
        if self.training:
            raise NotImplementedError('the split halves are an inference deployment (eval mode)')
        original_image_sizes = [tuple(img.shape[-2:]) for img in images]
        image_list, targets = self.transform(images, targets)
        x0 = self._stem.forward(to_nhwc(image_list.tensors, 4), False)
        if self.layer1_encoder.ext_classifier is not None:          # reference :28-33: the filter may stop here
            go_on, _ = self.layer1_encoder.filter(attach(E.logical(x0), x0))
            if not go_on:
                return None
        zb = self._head.forward_part(x0, 'encoder')
        z = attach(E.logical(zb, self._head.layers[self._head.encoder_len - 1].cout), zb)
        if self.bottleneck_transformer is not None:
            z, _ = self.bottleneck_transformer(z, targets)
        return z, tuple(image_list.tensors.shape), image_list.image_sizes, original_image_sizes


class RcnnTail(nn.Module):
    def __init__(self, rcnn_model, bottleneck_transformer=None):
        super().__init__()
        self.bottleneck_transformer = bottleneck_transformer
        backbone = rcnn_model.backbone
        self.layer1_decoder = backbone.body['layer1'].decoder
        self._head = backbone.body['layer1'].head_engine()
        self.sub_backbone = backbone
        self.rpn, self.roi_heads, self.transform = rcnn_model.rpn, rcnn_model.roi_heads, rcnn_model.transform
        self.features_only = False

    def forward(self, z, tensors_shape, image_sizes, original_image_sizes, targets=None):
        if self.training:
            raise NotImplementedError('the split halves are an inference deployment (eval mode)')
        if self.bottleneck_transformer is not None:
            z, _ = self.bottleneck_transformer(z, targets)
        out1 = self._head.forward_part(to_nhwc(z), 'decoder')
        cur = attach(E.logical(out1), out1)
        body = self.sub_backbone.body
        features = OrderedDict([(body.return_layers['layer1'], cur)])
        for name in ('layer2', 'layer3', 'layer4'):
            body[name]._keep = False
            cur = body[name](cur)
            if name in body.return_layers:
                features[body.return_layers[name]] = cur
        features = self.sub_backbone.fpn(features)
        if self.features_only:
            return features
        # reference :186-196: proposals -> detections -> back to the original image frame (the RPN only needs the
        # padded batch SHAPE, which is why the reference ships `tensors_shape` across the link)
        image_list = ImageList(_ShapeOnly(tensors_shape), image_sizes)
        proposals, _ = self.rpn(image_list, features, targets)
        detections, _ = self.roi_heads(features, proposals, image_sizes, targets)
        return self.transform.postprocess(detections, image_sizes, original_image_sizes)


class _ShapeOnly(object):
    """stands in for the batched image tensor on the tail's side of the link: only its shape exists there"""

    def __init__(self, shape):
        self.shape = tuple(shape)


def split_rcnn_model(model, quantization):
    """reference :215-221; ``quantization``: None, 8 or 16 (bits of the bottleneck codec)."""
    encoder_transformer = None if quantization is None else Compose([Quantizer(num_bits=quantization)])
    decoder_transformer = None if quantization is None else Compose([Dequantizer(num_bits=quantization)])
    return RcnnHead(model, encoder_transformer), RcnnTail(model, decoder_transformer)
