"""Bottleneck transformers (mirror of the reference's src/structure/transformer.py:131-174).

The reference applies them ONLY at evaluation time with ``-transform_bottleneck`` (src/models/mimic/base.py:54-57;
mimic_runner.py:90 disables them while distilling).  ``Quantizer`` / ``Dequantizer`` run as fused HIP kernels
(global min/max -> affine uint8 quantise; dequantise) on the NHWC bottleneck buffer; the 16-bit variants are a
half round trip.  ``JpegCompressor`` / ``JpegDecompressor`` (:94-128: the uint8 bottleneck image through a JPEG file)
and ``DataLogger`` (:58-91: pickled sizes of the bottleneck for the reference's cost analysis) are host-side file /
bookkeeping utilities: the quantisation they contain runs on the same HIP kernel, the rest is PIL / pickle as in the
reference.

Input pipeline (reference :32-55): ``ToTensor`` keeps the decoded uint8 image as a
``DecodedImage`` instead of materialising a float CHW copy on the host, ``RandomHorizontalFlip`` only flips the
(tiny) targets and marks the image; the /255, the flip and the model's normalise/resize/pad then run as ONE HIP
kernel inside ``CustomRCNNTransform`` (hnd_transform_image_u8): 4x fewer host->device bytes, no host float work.
"""
import os
import random

import numpy as np
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


class DecodedImage(object):
    """A decoded image waiting for the device: uint8 [H, W, 3] (``hwc``) or [3, H, W], plus a pending horizontal
    flip.  Quacks like the float CHW tensor the reference's dataset yields where the host looks at it
    (``.shape``, ``.to(device)``, ``.device``); the float values only ever exist inside the transform kernel."""

    def __init__(self, data, hwc, flip=False):
        assert data.dtype == torch.uint8 and data.dim() == 3 and data.shape[2 if hwc else 0] == 3
        self.data, self.hwc, self.flip = data, hwc, flip

    @property
    def shape(self):
        d = self.data.shape
        return torch.Size((3, d[0], d[1]) if self.hwc else tuple(d))

    @property
    def device(self):
        return self.data.device

    @property
    def is_cuda(self):
        return self.data.is_cuda

    def dim(self):
        return 3

    def to(self, *args, **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if k != 'dtype'}
        args = tuple(a for a in args if not isinstance(a, torch.dtype))
        return DecodedImage(self.data.to(*args, **kwargs), self.hwc, self.flip)

    def pin_memory(self, *args, **kwargs):
        """DataLoader(pin_memory=True) calls this on batch elements that have it"""
        return DecodedImage(self.data.pin_memory(*args, **kwargs), self.hwc, self.flip)

    def float_chw(self):
        """What the reference's ToTensor (+ flip) would have produced -- for checks, not used by the product path."""
        x = self.data.permute(2, 0, 1) if self.hwc else self.data
        x = x.float() / 255
        return x.flip(-1) if self.flip else x


def flip_coco_person_keypoints(kps, width):
    """COCO left/right joint swap + x mirror, zeroing invisible joints (reference :12-20)."""
    order = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
    out = kps[:, order]
    out[..., 0] = width - out[..., 0]
    out[out[..., 2] == 0] = 0
    return out


class RandomHorizontalFlip(object):
    """reference :32-49.  A DecodedImage is flipped lazily (inside the device kernel); tensors are flipped here."""

    def __init__(self, prob, rng=None):
        self.prob = prob
        self.rng = rng          # None: Python's global ``random`` like the reference; a random.Random: a stream of its own

    def __call__(self, image, target):
        if (self.rng or random).random() >= self.prob:
            return image, target
        width = image.shape[-1]
        if isinstance(image, DecodedImage):
            image = DecodedImage(image.data, image.hwc, not image.flip)
        else:
            image = image.flip(-1)
        boxes = target['boxes']
        boxes[:, [0, 2]] = width - boxes[:, [2, 0]]
        target['boxes'] = boxes
        if 'masks' in target:
            target['masks'] = target['masks'].flip(-1)
        if 'keypoints' in target:
            target['keypoints'] = flip_coco_person_keypoints(target['keypoints'], width)
        return image, target


class ToTensor(object):
    """reference :52-55 (functional.to_tensor).  With ``decoded`` (default) PIL images / uint8 arrays stay uint8
    (DecodedImage: the /255 happens inside the device transform kernel); ``decoded=False`` returns the float CHW
    tensor on the host exactly like the reference."""

    def __init__(self, decoded=True):
        self.decoded = decoded

    def __call__(self, image, target):
        image, target = self._defer(image, target)
        if not self.decoded and isinstance(image, DecodedImage):
            image = image.float_chw().contiguous()
        return image, target

    def _defer(self, image, target):
        if isinstance(image, DecodedImage) or (torch.is_tensor(image) and image.dtype != torch.uint8):
            return image, target
        if torch.is_tensor(image):          # uint8 tensor: [H, W, 3] unless it is unambiguously [3, H, W]
            return DecodedImage(image.contiguous(), hwc=image.shape[0] != 3), target
        arr = np.asarray(image)
        if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
            raise TypeError('ToTensor expects an RGB uint8 image, got %s %s' % (arr.dtype, arr.shape))
        return DecodedImage(torch.from_numpy(np.array(arr, order="C")), hwc=True), target       # copy: PIL memory is read-only


class DataLogger(object):
    """reference :58-91: records, per call, the pickled size (KB) of the bottleneck tensor as it is, as int16
    (``z.short()``, the reference's "fp16" figure) and quantised to ``num_bits``, plus its C / H / W; passes z through.
    A bottleneck transformer for analysis runs (``models.get_model(..., bottleneck_transformer=DataLogger())``)."""

    def __init__(self, num_bits=8):
        self.num_bits4quant = num_bits
        self.data_size_list, self.fp16_data_size_list = [], []
        self.quantized_data_size_list, self.tensor_shape_list = [], []

    def get_data(self):
        return (self.data_size_list.copy(), self.fp16_data_size_list, self.quantized_data_size_list.copy(),
                self.tensor_shape_list.copy())

    def clear(self):
        for lst in (self.data_size_list, self.fp16_data_size_list, self.quantized_data_size_list,
                    self.tensor_shape_list):
            lst.clear()

    def __call__(self, z, target):
        from ..myutils.common import file_util
        if z is None:
            sizes = (0.0, 0.0, 0.0)
        elif not isinstance(z, torch.Tensor):
            sizes = (file_util.get_binary_object_size(z), None, None)
        else:
            plain = z.detach().cpu().contiguous()               # logical NCHW values, as the reference pickles them
            qz, _ = Quantizer(self.num_bits4quant)(z, None) if self.num_bits4quant == 8 else (plain.half(), None)
            if isinstance(qz, QuantizedTensor):                  # what crosses the link: uint8 tensor + scale + zero point
                qz = _HostQuantized(qz.tensor.cpu().contiguous(), float(qz.scale), int(qz.zero_point))
            sizes = (file_util.get_binary_object_size(plain), file_util.get_binary_object_size(plain.short()),
                     file_util.get_binary_object_size(qz))
        self.data_size_list.append(sizes[0])
        self.fp16_data_size_list.append(sizes[1])
        self.quantized_data_size_list.append(sizes[2])
        self.tensor_shape_list.append([0, 0, 0] if z is None else [z.shape[1], z.shape[2], z.shape[3]])
        return z, target


class _HostQuantized(object):
    """host copy of a quantised bottleneck (uint8 tensor, python scale / zero point): what myutils' QuantizedTensor
    pickles to"""

    def __init__(self, tensor, scale, zero_point):
        self.tensor, self.scale, self.zero_point = tensor, scale, zero_point


class JpegCompressor(object):
    """reference :94-114: a 3-channel bottleneck ([3, H, W] or [1, 3, H, W]) is quantised to uint8 (HIP kernel),
    written as a JPEG file and replaced by ``(file path, quantised tensor)``; anything else passes through"""

    def __init__(self, jpeg_quality=95, tmp_dir_path='./tmp/'):
        self.jpeg_quality, self.tmp_dir_path = jpeg_quality, tmp_dir_path
        os.makedirs(tmp_dir_path, exist_ok=True)
        self._quantizer = Quantizer(8)

    def save_image(self, z, output_file_path):
        from PIL import Image
        qz, _ = self._quantizer(z if z.dim() == 4 else z.unsqueeze(0), None)
        hwc = qz.tensor[0].permute(1, 2, 0).cpu().numpy()
        Image.fromarray(np.ascontiguousarray(hwc)).save(output_file_path, format='jpeg', quality=self.jpeg_quality)
        return qz

    def __call__(self, z, target):
        if isinstance(z, torch.Tensor) and ((z.dim() == 3 and z.shape[0] == 3) or
                                            (z.dim() == 4 and z.shape[0] == 1 and z.shape[1] == 3)):
            file_path = os.path.join(self.tmp_dir_path, '{}.jpg'.format(hash(z)))
            return (file_path, self.save_image(z, file_path)), target
        return z, target


class JpegDecompressor(object):
    """reference :117-128: ``(file path, quantised tensor)`` -> decoded RGB * 255 de-quantised with the stored scale /
    zero point: ``scale * (pixel - zero_point)``, as [1, 3, H, W] (target_dim 4) or [3, H, W], on the device the
    quantised tensor lives on"""

    def __init__(self, tmp_dir_path='./tmp/', target_dim=4):
        self.tmp_dir_path, self.target_dim = tmp_dir_path, target_dim

    def __call__(self, z, target):
        if isinstance(z, tuple) and isinstance(z[0], str):
            from PIL import Image
            qz = z[1]
            arr = np.array(Image.open(z[0]).convert("RGB"), dtype=np.uint8)          # a writable copy
            pix = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).float().div(255).to(qz.tensor.device)
            img = qz.scale * (pix * 255.0 - qz.zero_point)      # functional.to_tensor(img) * 255.0 in the reference
            return (img if self.target_dim != 4 else img.unsqueeze(0)), target
        return z, target


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


TRANSFORMER_CLASS_DICT = {'jpeg_compressor': JpegCompressor, 'jpeg_decompressor': JpegDecompressor,
                          'quantizer': Quantizer, 'dequantizer': Dequantizer}


def get_bottleneck_transformer(transformer_config):
    components = []
    for name in transformer_config['order']:
        if name not in TRANSFORMER_CLASS_DICT:
            raise KeyError('transformer `{}` is not expected'.format(name))
        components.append(TRANSFORMER_CLASS_DICT[name](**transformer_config['components'][name]['params']))
    return Compose(components) if components else None
