"""Bottleneck transformers (mirror of the reference's src/structure/transformer.py:131-174).

The reference applies them ONLY at evaluation time with ``-transform_bottleneck`` (src/models/mimic/base.py:54-57;
mimic_runner.py:90 disables them while distilling).  ``Quantizer`` / ``Dequantizer`` run as fused HIP kernels
(global min/max -> affine uint8 quantise; dequantise) on the NHWC bottleneck buffer; the 16-bit variants are a
half round trip.  JPEG codecs and the DataLogger of the reference are deployment/analysis tools outside the path.
"""
import torch

from .. import ops
from ..hipnn import attach, to_nhwc


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


class DataLogger(object):
    def __init__(self, num_bits=8):
        self.num_bits4quant = num_bits


class QuantizedTensor(object):
    """myutils tensor_util.QuantizedTensor: .tensor (uint8, logical NCHW), .scale, .zero_point.  scale / zero_point
    are 0-dim DEVICE tensors (views of the kernel's qparams) so no host sync is forced; int()/float() work."""

    def __init__(self, tensor, scale, zero_point, qparams=None, origin=None, channels=None):
        self.tensor, self.scale, self.zero_point = tensor, scale, zero_point
        self.qparams, self.origin, self.channels = qparams, origin, channels


class Quantizer(object):
    def __init__(self, num_bits=8):
        if num_bits not in (8, 16):
            raise NotImplementedError('HIP bottleneck codec implements 8- and 16-bit quantisation')
        self.num_bits = num_bits
        self._bufs = {}

    def __call__(self, z, target):
        buf = to_nhwc(z)                    # [N, H, W, cs] fp32 bottleneck (pad channels are 0)
        c = z.shape[1]
        if self.num_bits == 16:             # z.half(): stored back as the fp32 value of the half
            ops.roundtrip_f16(buf)
            return z, target
        key = (tuple(buf.shape), buf.device)
        if key not in self._bufs:
            self._bufs[key] = (torch.empty(buf.shape, dtype=torch.uint8, device=buf.device),
                               torch.empty(4, dtype=torch.float32, device=buf.device),
                               torch.empty(ops.minmax_scratch_elems(), dtype=torch.float32, device=buf.device))
        q, qparams, scratch = self._bufs[key]
        ops.quantize_u8(buf, c, q, qparams, scratch)
        qt = q[..., :c].permute(0, 3, 1, 2)
        return QuantizedTensor(attach(qt, q), qparams[2], qparams[3], qparams, origin=buf, channels=c), target


class Dequantizer(object):
    def __init__(self, num_bits=8):
        self.num_bits = num_bits

    def __call__(self, qz, target):
        if self.num_bits == 16 or not isinstance(qz, QuantizedTensor):
            return qz, target
        q = qz.tensor._hnd
        out = qz.origin if qz.origin is not None else torch.empty(q.shape, dtype=torch.float32, device=q.device)
        ops.dequantize_u8(q, qz.qparams, out, qz.channels)
        z = out[..., :qz.channels].permute(0, 3, 1, 2)
        return attach(z, out), target


TRANSFORMER_CLASS_DICT = {'quantizer': Quantizer, 'dequantizer': Dequantizer}


def get_bottleneck_transformer(transformer_config):
    components = []
    for name in transformer_config['order']:
        if name not in TRANSFORMER_CLASS_DICT:
            raise KeyError('transformer `{}` is not expected'.format(name))
        components.append(TRANSFORMER_CLASS_DICT[name](**transformer_config['components'][name]['params']))
    return Compose(components) if components else None
