"""Execution engines behind the nn.Module tree: prebuilt launch plans over libhnd_hip.so.

Each engine owns, per input geometry, its activation buffers (torch-ROCm tensors, NHWC) and a
list of prebuilt launches that is replayed every step.  Nothing here computes with torch ops.

  TransformEngine   CustomRCNNTransform.forward          (reference src/models/org/rcnn.py:65-82)
  StemEngine        conv1 -> FrozenBN -> ReLU -> maxpool (src/models/custom/resnet.py:26-30,96-99)
  HeadEngine        Bottleneck4LargeResNet, train-mode BN (src/models/mimic/resnet_layer.py:40-70)
  FrozenLayerEngine nn.Sequential of torchvision Bottleneck with FrozenBatchNorm2d (rcnn.py:391-395)
  FpnEngine         FeaturePyramidNetwork + LastLevelMaxPool (rcnn.py:399-414)

Backward is a hand-written plan as well (no autograd graph): dgrad through the frozen layers with
the ReLU mask / FrozenBN scale / residual fan-in fused into the conv kernel, train-BN backward and
dgrad+wgrad for the head, maxpool/ReLU/FBN backward and wgrad for the stem.
"""
import math
import os
from collections import OrderedDict

import torch

from . import ops

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
WINOGRAD = int(os.environ.get('HND_WINOGRAD', '4'))      # output tile of the Winograd 3x3 path: 4, 2, or 0 = off
WINOGRAD_HEAD = True     # F(4x4,2x2) / F(6x6,2x2) for the deep head convs (settled in round 2; was HND_WINOGRAD_HEAD)
PROFILE = {'enabled': False, 'records': []}     # bench.py: per-launch HIP events on the launch stream
# DistillationBox sets 'stream' while it runs teacher + student: their feature pyramids (whose outputs the
# distillation criterion never reads) are then issued on that stream and overlap the backward pass
DEFER_FPN = {'stream': None}


def _run(launch, tag=None):
    if PROFILE['enabled'] and tag is not None:          # every tagged plan entry, incl. the flop-less Winograd transforms
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch.run()
        e1.record()
        PROFILE['records'].append((tag or type(launch).__name__, launch, e0, e1))
    else:
        launch.run()


def version_of(tensors):
    return tuple(t._version for t in tensors)


def weight_version(w):
    """staleness key of a packed / transformed copy of parameter `w`.  The fused optimizers update parameters through
    raw pointers (optim.py -> hnd_adam_step_flat / hnd_sgd_step_flat), which never bumps ``_version``; they advance
    ``ops.PARAM_EPOCH`` instead, so a train-then-eval sequence in one process repacks trainable weights."""
    return (w._version, w.data_ptr(), ops.PARAM_EPOCH[0] if w.requires_grad else -1)


class Buffers(object):
    """Named persistent device buffers of one engine (re-allocated only when a shape changes)."""

    def __init__(self, device):
        self.device = device
        self.t = {}

    def get(self, name, shape, dtype=torch.float32):
        shape = tuple(int(s) for s in shape)
        cur = self.t.get(name)
        if cur is None or tuple(cur.shape) != shape or cur.dtype != dtype:
            cur = torch.empty(shape, dtype=dtype, device=self.device)
            self.t[name] = cur
        return cur

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.t.values())


def logical(t_nhwc, c=None):
    """NHWC buffer -> logical NCHW view (what forward hooks / callers observe)."""
    if c is not None and c != t_nhwc.shape[3]:
        t_nhwc = t_nhwc[..., :c]
    return t_nhwc.permute(0, 3, 1, 2)


def physical(t_nchw):
    """logical NCHW tensor backed by an NHWC buffer -> the NHWC buffer view; else a contiguous NHWC copy is
    refused (callers must hand over channels_last device tensors produced by this package)."""
    p = t_nchw.permute(0, 2, 3, 1)
    if not p.is_contiguous():
        raise RuntimeError('expected a channels_last tensor produced by the HIP path; got strides %s'
                           % (tuple(t_nchw.stride()),))
    return p


class FrozenAffine(object):
    """FrozenBatchNorm2d folded to per-channel (scale, shift) by hnd_fbn_fold; refreshed when buffers change."""

    def __init__(self, bn):
        self.bn = bn
        self.ver = None
        self.scale = self.shift = None

    def get(self):
        ts = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        ver = version_of(ts) + tuple(t.data_ptr() for t in ts)
        if ver != self.ver:
            self.scale, self.shift = ops.fbn_fold(*ts, eps=0.0)
            self.ver = ver
        return self.scale, self.shift


class WeightCache(object):
    """Packed GEMM operands of one conv weight; frozen weights are packed once, trainable ones every step."""

    def __init__(self, weight):
        self.weight = weight
        self.packs = {}
        self.ver = None

    def get(self, transposed=False, chan_pad=None, taps=None, kscale=None):
        """the PackedWeight for this layout; its buffer address is stable for the life of the cache (prebuilt
        launch descriptors point at it), refresh() re-runs the pack kernel into the same buffer.
        kscale: per-channel scale folded into the K operand (ops.pack_weights); a NEW scale tensor (the FrozenBN fold
        was refreshed) re-packs into the same buffer."""
        key = (transposed, chan_pad, taps, kscale is not None)
        pk = self.packs.get(key)
        if pk is None:
            pk = ops.pack_weights(self.weight.detach(), transposed, chan_pad, taps, kscale=kscale)
            self.packs[key] = pk
        elif kscale is not None and pk.kscale is not kscale:
            pk.kscale = kscale
            pk.repack()
        return pk

    def refresh(self, force=False):
        """re-run the pack kernels if the parameter changed (or always, for trainable weights)."""
        ver = weight_version(self.weight)
        if force or ver != self.ver:
            for pk in self.packs.values():
                pk.src = self.weight.detach()
                pk.repack()
            self.ver = ver


class WinoCache(object):
    """Winograd-domain weights (forward and transposed) of one frozen 3x3 conv, transformed once per version."""

    def __init__(self, weight, tile):
        self.weight, self.tile, self.packs, self.ver = weight, tile, {}, None

    def get(self, dgrad=False, tile=None):
        """tile: the output tile the launch will use (wino_tile_for picks it per geometry); default = the cache's own"""
        tile = tile or self.tile
        ww = self.packs.get((dgrad, tile))
        if ww is None:
            ww = self.packs[(dgrad, tile)] = ops.WinoWeights(self.weight.detach(), dgrad, tile)
            self.ver = weight_version(self.weight)
        return ww

    def refresh(self):
        ver = weight_version(self.weight)
        if ver != self.ver:
            for ww in self.packs.values():
                ww.src = self.weight.detach()
                ww.repack()
            self.ver = ver


class Wino2Cache(object):
    """F(4x4,2x2) weights (forward / transposed) of one head conv; trainable, so re-transformed every step."""

    def __init__(self, weight):
        self.weight, self.packs, self.ver = weight, {}, None

    def get(self, dgrad=False, tile=4):
        ww = self.packs.get((dgrad, tile))
        if ww is None:
            ww = self.packs[(dgrad, tile)] = ops.Wino2Weights(self.weight.detach(), dgrad, tile)
        return ww

    def refresh(self, force=False):
        ver = weight_version(self.weight)
        if force or ver != self.ver:
            for ww in self.packs.values():
                ww.src = self.weight.detach()
                ww.repack()
            self.ver = ver


FOLD_DGRAD_SCALE = True     # (False: FrozenBN scale as a launch prologue; settled in round 3, was HND_FOLD_DGRAD_SCALE)
# ReLU masks of the Bottleneck outputs as nibbles (one byte per pixel and 4 channels, written by the forward epilogue that
# stores the output) instead of re-reading the fp32 activation in the conv1 data gradients -- HBM-bound launches (out +
# residual + mask at K = 128): 1/16 of the mask's bytes.  Same decisions (x > 0), same bits.  0: fp32 masks (A/B, tests)
MASK_BITS = True            # (settled in round 4; was HND_MASK_BITS)
# BatchNorm backward "apply" of the two deep decoder convs fused into the two transforms that consume dy (ops.wino26_bnbwd_step)
FUSE_BNBWD = True           # (settled in round 4; was HND_FUSE_BNBWD)
WINO_WGRAD_OWN_V = True     # (settled in round 4; was HND_WINO_WGRAD_OWN_V)


def process_owns_device(verbose=[True]):
    """False when several local ranks time-share this process's GPU: extra HIP streams then make a step ~50x slower
    (distillation/tool.py), so the optional side streams stay off there.  Decided from what is actually shared:
      1. a launcher that narrowed the visible-device list per rank (SLURM-style: HIP_/ROCR_/CUDA_VISIBLE_DEVICES names
         exactly one device while LOCAL_WORLD_SIZE > 1) gave this rank a GPU of its own -> owned;
      2. LOCAL_WORLD_SIZE known: owned iff it does not exceed the visible devices;
      3. only WORLD_SIZE known (no LOCAL_WORLD_SIZE): it may span nodes, so it is used only when it fits the visible
         devices; a WORLD_SIZE larger than this node's GPUs says nothing about sharing -> assumed owned.
    ``HND_SHARED_DEVICE=1`` (bench.py --share_device, the one-GPU plumbing tests) declares the device shared whatever the
    visible-device list says.  The rule that fired is printed once per process."""
    def env_int(name):
        try:
            return int(os.environ[name])
        except (KeyError, ValueError):
            return None
    devices = max(torch.cuda.device_count() if torch.cuda.is_available() else 1, 1)
    local_world, world = env_int('LOCAL_WORLD_SIZE'), env_int('WORLD_SIZE')
    narrowed = [v for v in (os.environ.get(k) for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))
                if v is not None and len([d for d in v.split(',') if d.strip() != '']) == 1]
    if os.environ.get('HND_SHARED_DEVICE', '0') != '0':
        owns, rule = False, 'HND_SHARED_DEVICE=1'
    elif local_world is not None and local_world > 1 and narrowed and devices == 1:
        owns, rule = True, 'visible devices narrowed to one per rank (LOCAL_WORLD_SIZE=%d)' % local_world
    elif local_world is not None:
        owns, rule = local_world <= devices, 'LOCAL_WORLD_SIZE=%d vs %d visible device(s)' % (local_world, devices)
    elif world is not None and world <= devices:
        owns, rule = True, 'WORLD_SIZE=%d <= %d visible device(s)' % (world, devices)
    elif world is not None:
        owns, rule = True, 'WORLD_SIZE=%d spans nodes (no LOCAL_WORLD_SIZE): assuming one rank per GPU' % world
    else:
        owns, rule = True, 'single process'
    if verbose[0] and (local_world or world or 1) > 1:
        verbose[0] = False
        print('engine.process_owns_device: %s -> side streams %s' % (rule, 'on' if owns else 'off'))
    return owns


def wgrad_stream_on():
    """the head's weight-gradient chains (independent of the data-gradient chain given dy) on a stream of their own:
    HND_WGRAD_STREAM=0/1 decides when set, else on when this process has its GPU to itself"""
    env = os.environ.get('HND_WGRAD_STREAM')
    return (env != '0') if env is not None else process_owns_device()


# BatchNorm backward "reduce" of a head layer folded into the output transform of the data gradient that produces its g
FOLD_BNBWD_REDUCE = True    # (settled in round 4; was HND_FOLD_BNBWD_REDUCE)
WINOGRAD6 = os.environ.get('HND_WINOGRAD6', '1') != '0'       # F(6x6,3x3) on maps large enough (wino_tile_for)
# Which of the STUDENT's forward launches the bf16x3 emulation may take (the teacher, the backward pass and the loss-dead
# pyramid always may): the student's forward pass makes the ReLU / BatchNorm decisions every gradient hangs on
BX3_STUDENT_FWD = os.environ.get('HND_BX3_STUDENT_FWD', '1') != '0'
_hf = os.environ.get('HND_BX3_HEAD_FWD', '1')        # '0' none, '1' all, or the head conv indices, e.g. '0,1,5'
BX3_HEAD_FWD = set(range(8)) if _hf == '1' else (set() if _hf == '0' else set(int(v) for v in _hf.split(',')))


def use_winograd(cin, cout, stride):
    """Winograd output tile (0 = direct) for a 3x3 conv.  It pays when the (tile+2)^2 GEMMs are deep enough to run
    at MFMA rate and the inflated transformed tensors stay cheap next to them (tools/bench_wino.py, batch 16):
    F(4x4,3x3) x2.2-2.8 over the direct kernel for 256/512 channels and x1.7 for 128; F(2x2,3x3) x1.3-2.0 for
    256/512 and x1.0 for 128."""
    if WINOGRAD not in (2, 4) or stride != 1 or cin % 32 != 0 or cout % 4 != 0:
        return 0
    # with F(6x6,3x3) available even the 64-channel convs of the teacher's layer1 gain (x1.20 at 200x336,
    # profiles/r03_bench_wino.txt; F(4x4): x1.06): their 64 component GEMMs are K = 64 deep, HBM-bound like the transforms
    floor = (64 if WINOGRAD6 else 128) if WINOGRAD == 4 else 256
    return WINOGRAD if min(cin, cout) >= floor else 0


WINOGRAD6_MIN_TILES = 32      # 6x6 tiles per IMAGE below which F(4x4,.) stays (border waste of the 6x6 tiling)


def wino_tile_for(tile, n, h, w):
    """F(6x6,3x3) instead of F(4x4,3x3) where the map is large enough: 64 products per 36 outputs instead of 36 per
    16 (21 % fewer GEMM flops) and transformed tensors of 1.78x instead of 2.25x the activation (21 % fewer transform
    bytes), at twice F(4x4)'s -- still ~5e-6 -- fp32 error (measured: tools/bench_wino.py, profiles/r03_bench_wino.txt:
    x1.12-1.19 over F(4x4) from 200x336 down to 25x42 at 512 channels).  Tiny maps keep 4x4 tiles (border waste of
    the 6x6 tiling).  The choice depends on the map size only, never on the batch: the arithmetic of one image is the
    same alone and inside a batch (tests: batch-16 maps == batch-1 maps bit for bit)."""
    if tile != 4 or not WINOGRAD6:
        return tile
    return 6 if ((h + 5) // 6) * ((w + 5) // 6) >= WINOGRAD6_MIN_TILES else 4


WINOGRAD2_6 = True          # (settled in round 3; was HND_WINOGRAD2_6)


def wino2_tile_for(oh, ow):
    """F(6x6,2x2) (49 products per 36 outputs) instead of F(4x4,2x2) (25 per 16) for the deep head convs on maps of at
    least WINOGRAD6_MIN_TILES 6x6 tiles per image: 12.9 % fewer GEMM flops, transformed tensors 1.36x instead of 1.56x
    the activation; fp32 error 4.1e-6 vs 1.7e-6 relative L2.  Map size only, never the batch (see wino_tile_for)."""
    if not WINOGRAD2_6:
        return 4
    return 6 if ((oh + 5) // 6) * ((ow + 5) // 6) >= WINOGRAD6_MIN_TILES else 4


# =========================================================================================== transform
_TRANSFORMS = {}


def shared_transform(mean, std, device):
    """teacher and student normalise identically (rcnn.py:222-226), so they share one engine: the second
    model's transform of the same image list is a cache hit instead of a second pass."""
    key = (tuple(float(m) for m in mean), tuple(float(s) for s in std), str(device))
    if key not in _TRANSFORMS:
        _TRANSFORMS[key] = TransformEngine(mean, std)
    return _TRANSFORMS[key]


_SCOPE = {'id': None, 'next': 1}


def transform_scope_begin():
    """DistillationBox.forward brackets its teacher+student calls with a scope: inside it the image list is
    alive and unchanged, so the second model's identical transform is served from the first one's batch."""
    _SCOPE['id'] = _SCOPE['next']
    _SCOPE['next'] += 1


def transform_scope_end():
    _SCOPE['id'] = None


def _image_key(im):
    t = im.data if hasattr(im, 'hwc') else im
    return (id(im), t.data_ptr(), t._version, bool(getattr(im, 'flip', False)))


class TransformEngine(object):
    def __init__(self, mean, std):
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self.bufs = None
        self.last_key = None

    def run(self, images, sizes, max_size):
        """images: list of CHW device tensors (or uint8 DecodedImage of structure.transformer: /255 and the
        pending flip are fused into the same kernel); sizes: per-image target min side.
        Returns (NHWC4 batch, image_sizes)."""
        dev = images[0].device
        if self.bufs is None:
            self.bufs = Buffers(dev)
        plans = []
        for img, size in zip(images, sizes):
            if img.dim() != 3 or img.shape[0] != 3:
                raise ValueError('images is expected to be a list of 3d tensors of shape [3, H, W], got %s'
                                 % (tuple(img.shape),))
            h, w = int(img.shape[1]), int(img.shape[2])
            scale = float(size) / float(min(h, w))              # rcnn.py:41-43
            if float(max(h, w)) * scale > max_size:
                scale = float(max_size) / float(max(h, w))
            plans.append((h, w, scale, ops.interp_out_size(h, scale), ops.interp_out_size(w, scale)))
        hp = int(math.ceil(max(p[3] for p in plans) / 32.0) * 32)
        wp = int(math.ceil(max(p[4] for p in plans) / 32.0) * 32)
        key = (_SCOPE['id'],) + tuple(_image_key(im) + p for im, p in zip(images, plans))
        batch = self.bufs.get('batch', (len(images), hp, wp, 4))
        if _SCOPE['id'] is None or key != self.last_key:
            items, keep = [], []
            for img, (h, w, scale, oh, ow) in zip(images, plans):
                if hasattr(img, 'hwc'):         # DecodedImage: /255 and the pending flip are fused into the kernel
                    src = img.data if img.data.is_contiguous() else img.data.contiguous()
                    items.append((src, True, img.hwc, img.flip, oh, ow, 1.0 / scale, 1.0 / scale))
                else:
                    src = img if (img.is_contiguous() and img.dtype == torch.float32) else img.float().contiguous()
                    items.append((src, False, False, False, oh, ow, 1.0 / scale, 1.0 / scale))
                keep.append(src)
            ops.transform_images(items, batch, self.mean, self.std)          # one launch for the whole batch
            self.last_key = key
        self.last_scales = [p[2] for p in plans]
        return batch, [(p[3], p[4]) for p in plans]


# =========================================================================================== stem
class StemEngine(object):
    def __init__(self, conv1, bn1):
        self.conv1, self.fbn = conv1, FrozenAffine(bn1)
        self.wc = WeightCache(conv1.weight)
        self.bufs = None
        self.plan_key = None
        self.flops_fwd = self.flops_bwd = 0

    def forward(self, x4, keep):
        n, hp, wp, _ = x4.shape
        if self.bufs is None:
            self.bufs = Buffers(x4.device)
        scale, shift = self.fbn.get()
        pk = self.wc.get(False, 4)
        self.wc.refresh(force=self.conv1.weight.requires_grad)
        oh, ow = ops.conv_out_size(hp, 7, 2, 3), ops.conv_out_size(wp, 7, 2, 3)
        ph, pw_ = ops.conv_out_size(oh, 3, 2, 1), ops.conv_out_size(ow, 3, 2, 1)
        key = (x4.data_ptr(), tuple(x4.shape), scale.data_ptr())
        if key != self.plan_key:
            b = self.bufs
            self.a0 = b.get('a0', (n, oh, ow, 64))
            self.x0 = b.get('x0', (n, ph, pw_, 64))
            self.idx = b.get('idx', (n, ph, pw_, 64), torch.uint8)
            self.l_conv = ops.conv_forward(x4, pk, self.a0, 7, 2, 3, epi_scale=scale, epi_shift=shift, relu=True)
            self.flops_fwd = 2 * n * oh * ow * 64 * 49 * 3
            self.x4 = x4
            self.plan_key = key
            self.bwd = None
        _run(self.l_conv, 'stem.conv1')
        ops.maxpool_fwd(self.a0, self.x0, self.idx)
        return self.x0

    def backward(self, g_x0, dw):
        """g_x0: grad wrt the pooled stem output; dw: destination of d conv1.weight (or None when frozen)."""
        if dw is None:
            return
        scale, _ = self.fbn.get()
        key = (g_x0.data_ptr(), dw.data_ptr())
        if self.bwd is None or self.bwd[0] != key:
            dconv = self.bufs.get('dconv', self.a0.shape)
            wl = ops.conv_wgrad(self.x4, dconv, dw, 7, 2, 3)
            self.bwd = (key, dconv, wl)
            n, oh, ow, _ = self.a0.shape
            self.flops_bwd = 2 * n * oh * ow * 64 * 49 * 3
        _, dconv, wl = self.bwd
        ops.maxpool_bwd_relu_scale(g_x0, self.idx, self.a0, scale, dconv)
        _run(wl, 'stem.wgrad')


# =========================================================================================== frozen layers
class _Block(object):
    __slots__ = ('mod', 'stride', 'has_ds', 'planes', 'cin', 'w1', 'w2', 'w3', 'wd', 'f1', 'f2', 'f3', 'fd', 'wino')


class FrozenLayerEngine(object):
    """A run of torchvision Bottleneck blocks whose convs and FrozenBatchNorm2d are frozen."""

    def __init__(self, blocks, name):
        self.name = name
        self.blocks = []
        for m in blocks:
            b = _Block()
            b.mod, b.stride = m, m.stride
            b.has_ds = m.downsample is not None
            b.planes, b.cin = m.conv1.weight.shape[0], m.conv1.weight.shape[1]
            b.w1, b.w2, b.w3 = WeightCache(m.conv1.weight), WeightCache(m.conv2.weight), WeightCache(m.conv3.weight)
            b.f1, b.f2, b.f3 = FrozenAffine(m.bn1), FrozenAffine(m.bn2), FrozenAffine(m.bn3)
            b.wd = WeightCache(m.downsample[0].weight) if b.has_ds else None
            b.fd = FrozenAffine(m.downsample[1]) if b.has_ds else None
            tile = use_winograd(b.planes, b.planes, b.stride)
            b.wino = WinoCache(m.conv2.weight, tile) if tile else None
            self.blocks.append(b)
        self.bufs = None
        self.plan_key = None
        self.bwd_key = None
        self.flops_fwd = self.flops_bwd = 0
        # out_provider(shape) -> NHWC tensor: where the layer's output goes instead of a buffer of its own (SharedTrunk:
        # layer1 of teacher / student write straight into their half of the concatenated batch)
        self.out_provider = None
        # bslice = (lo, hi): backward runs over images lo..hi of the forward batch only (SharedTrunk: the student half)
        self.bslice = None

    def _b(self, t):
        return t if self.bslice is None else t[self.bslice[0]:self.bslice[1]]

    def bwd_out(self):
        """the layer output the backward pass sees (ReLU mask of the gradient that enters the NEXT layer's dgrad)"""
        return self._b(self.out)

    def bwd_out_bits(self):
        """... and its ReLU mask as nibbles (None when the forward did not produce them)"""
        bits = self.out_bits[-1] if getattr(self, 'out_bits', None) else None
        return None if bits is None else self._b(bits)

    def _check_frozen(self):
        for b in self.blocks:
            for p in b.mod.parameters():
                if p.requires_grad:
                    raise NotImplementedError(
                        '%s has trainable parameters: the HIP path implements the reference distillation '
                        'configs, which freeze layer2-4/FPN (yaml student_model.frozen_modules)' % self.name)

    def out_shape(self, x):
        n, h, w, _ = x.shape
        for b in self.blocks:
            h, w = ops.conv_out_size(h, 3, b.stride, 1), ops.conv_out_size(w, 3, b.stride, 1)
        return (n, h, w, self.blocks[-1].planes * 4)

    def prepare(self, x, keep):
        """refresh the operands and (re)build the launch plan for this input without launching anything; afterwards
        ``self.out`` is the buffer the next forward() will fill"""
        if self.bufs is None:
            self.bufs = Buffers(x.device)
        affs = [(b.f1.get(), b.f2.get(), b.f3.get(), b.fd.get() if b.has_ds else None) for b in self.blocks]
        for b in self.blocks:
            for wc in (b.w1, b.w2, b.w3, b.wd):
                if wc is not None and not (wc is b.w2 and b.wino is not None):
                    wc.get()
                    wc.refresh()
            if b.wino is not None:          # (its packs are made by the plans, for the tile the geometry picks)
                b.wino.refresh()
        self._out_buf = self.out_provider(self.out_shape(x)) if self.out_provider is not None else None
        key = (x.data_ptr(), tuple(x.shape), keep, getattr(self, 'for_backward', True),
               tuple(a[0][0].data_ptr() for a in affs), None if self._out_buf is None else self._out_buf.data_ptr())
        if key != self.plan_key:
            with ops.emulation_unless(not keep or BX3_STUDENT_FWD):
                self._build_forward(x, keep, affs)
            self.plan_key = key
            self.bwd_key = None
        return self.out

    def forward(self, x, keep):
        """x: NHWC input. keep=True retains every activation for backward (student), else buffers ping-pong."""
        self.prepare(x, keep)
        for l, tag in self.fwd:
            _run(l, tag)
        return self.out

    def _build_forward(self, x, keep, affs):
        self.fwd, self.acts, self.out_bits, self.a2_bits = [], [], [], []
        n = x.shape[0]
        cur = x
        flops = 0
        for i, (b, (a1f, a2f, a3f, adf)) in enumerate(zip(self.blocks, affs)):
            h, w = cur.shape[1], cur.shape[2]
            oh, ow = ops.conv_out_size(h, 3, b.stride, 1), ops.conv_out_size(w, 3, b.stride, 1)
            tagp = '%s.%d' % (self.name, i)
            sfx = str(i) if keep else str(i & 1)
            a1 = self.bufs.get('a1_' + sfx, (n, h, w, b.planes))
            a2 = self.bufs.get('a2_' + sfx, (n, oh, ow, b.planes))
            out = self.bufs.get('out_' + sfx, (n, oh, ow, b.planes * 4))
            if i == len(self.blocks) - 1 and self._out_buf is not None:
                assert tuple(self._out_buf.shape) == tuple(out.shape), (self._out_buf.shape, out.shape)
                out = self._out_buf
            self.fwd.append((ops.conv_forward(cur, b.w1.get(), a1, 1, 1, 0, epi_scale=a1f[0], epi_shift=a1f[1],
                                              relu=True), tagp + '.conv1'))
            # with the bf16x3 emulation (the default): [a2 > 0] as nibbles for conv3's data gradient -- the emulation kernel's
            # masked build reads mask BYTES (a second fp32 row set does not fit its registers).  Since round 6 they come out of
            # the launch that stores a2 (the Winograd output transform / the conv epilogue), not a pass of their own
            a2b = None
            if (keep and getattr(self, 'for_backward', True) and ops.bx3_on() and MASK_BITS
                    and b.planes % 128 == 0):
                a2b = self.bufs.get('a2bits_' + sfx, tuple(a2.shape[:3]) + (a2.shape[3] // 4,), torch.uint8)
            bits_step = a2b is not None
            if b.wino is not None:      # stride-1 3x3, >= 256 channels: Winograd F(2x2,3x3)
                tile = wino_tile_for(b.wino.tile, n, h, w)
                v, m = self._wino_scratch(n, h, w, b.planes, b.planes, tile)
                in_transform = a2b is not None and tile in (4, 6)
                self.fwd += ops.WinoConv(a1, b.wino.get(False, tile), a2, v, m, epi_scale=a2f[0], epi_shift=a2f[1],
                                         relu=True, mask_out=a2b if in_transform else None).launches(tagp + '.conv2')
                bits_step = bits_step and not in_transform
            else:
                self.fwd.append((ops.conv_forward(a1, b.w2.get(), a2, 3, b.stride, 1, epi_scale=a2f[0],
                                                  epi_shift=a2f[1], relu=True, mask_out=a2b), tagp + '.conv2'))
                bits_step = False
            if bits_step:               # (F(2x2,3x3): its output transform owns whole pixels per thread, not channel pairs)
                self.fwd.append((ops._Step(lambda s, a2=a2, a2b=a2b: ops.relu_mask_nibbles(a2, a2b, s), 'relu_mask_nibbles',
                                           a2.numel() * 4 + a2b.numel()), tagp + '.conv2.bits'))
            self.a2_bits.append(a2b)
            if b.has_ds:
                ds = self.bufs.get('ds_' + sfx, (n, oh, ow, b.planes * 4))
                self.fwd.append((ops.conv_forward(cur, b.wd.get(), ds, 1, b.stride, 0, epi_scale=adf[0],
                                                  epi_shift=adf[1]), tagp + '.downsample'))
                ident = ds
                flops += 2 * n * oh * ow * b.planes * 4 * b.cin
            else:
                ident = cur
            # (keep = the student: the backward of the NEXT block / layer applies [out > 0] -- as nibbles, see MASK_BITS)
            bits = self.bufs.get('bits_' + sfx, tuple(out.shape[:3]) + (out.shape[3] // 4,), torch.uint8) \
                if (keep and MASK_BITS and out.shape[3] % 128 == 0) else None
            self.fwd.append((ops.conv_forward(a2, b.w3.get(), out, 1, 1, 0, epi_scale=a3f[0], epi_shift=a3f[1],
                                              res1=ident, relu=True, mask_out=bits), tagp + '.conv3'))
            self.out_bits.append(bits)
            flops += 2 * n * (h * w * b.planes * b.cin + oh * ow * b.planes * b.planes * 9
                              + oh * ow * b.planes * 4 * b.planes)
            self.acts.append((cur, a1, a2, out))
            cur = out
        self.out = cur
        self.flops_fwd = flops

    def _wino_scratch(self, n, h, w, cin, cout, tile=None):
        """V / M scratch of the Winograd launches of this engine (they run one after another on one stream)."""
        nv, nm = ops.WinoConv.scratch_elems(n, h, w, cin, cout, tile or WINOGRAD)
        self._wino_need = (max(nv, getattr(self, '_wino_need', (0, 0))[0]), max(nm, getattr(self, '_wino_need', (0, 0))[1]))
        return self.bufs.get('wino_v', (self._wino_need[0],)), self.bufs.get('wino_m', (self._wino_need[1],))

    # ---- backward: self.g_out holds the gradient w.r.t. this layer's output, already masked by out > 0
    def block_out(self, i):
        """output of Bottleneck i as the backward pass sees it (forward hooks on ``layerN.i``)"""
        return self._b(self.acts[i][3])

    def grad_out_buffer(self, top_block=None):
        """where the (ReLU-masked) gradient w.r.t. the layer output -- or, with top_block, w.r.t. the output of that
        inner Bottleneck, when the highest loss term sits there -- is written before backward()"""
        if top_block is None or top_block == len(self.blocks) - 1:
            self.g_out = self.bufs.get('g_out', self.bwd_out().shape)
        else:
            self.g_out = self.bufs.get('g_out_blk%d' % top_block, self.block_out(top_block).shape)
        return self.g_out

    def backward(self, dst, dst_mask, res2, top_block=None, block_grads=None, dst_mask_bits=None):
        """Propagate self.g_out to `dst` (grad w.r.t. this layer's input):
        dst = [dst_mask > 0] * (dgrad + res2).  res2 (the loss gradient of the previous layer) may be None.
        top_block: the gradient enters at the output of that Bottleneck (later blocks carry no loss: skipped);
        block_grads: {i: unmasked loss gradient w.r.t. the output of Bottleneck i}, added where the gradient of that
        output is formed (and masked by its ReLU there)."""
        self._check_frozen()
        block_grads = block_grads or {}
        key = (dst.data_ptr(), dst_mask.data_ptr(), None if res2 is None else res2.data_ptr(), self.g_out.data_ptr(),
               self.bslice, top_block, tuple(sorted((i, t.data_ptr()) for i, t in block_grads.items())),
               None if dst_mask_bits is None else dst_mask_bits.data_ptr())
        if key != self.bwd_key:
            self._build_backward(dst, dst_mask, res2, top_block, block_grads, dst_mask_bits)
            self.bwd_key = key
        for l, tag in self.bwd:
            _run(l, tag)

    def _build_backward(self, dst, dst_mask, res2, top_block=None, block_grads=None, dst_mask_bits=None):
        self.bwd = []
        flops = 0
        g = self.g_out
        nb = len(self.blocks) if top_block is None else top_block + 1
        block_grads = block_grads or {}

        def fold(scale):
            # the FrozenBN scale between a conv and the gradient reaching it: folded into the transposed weights
            # (one-time, the weights are frozen) instead of a prologue on every data-gradient launch
            return {'fold_scale': scale} if FOLD_DGRAD_SCALE else {'pro_scale': scale}

        for i in range(nb - 1, -1, -1):
            b = self.blocks[i]
            x_in, a1, a2, out = (self._b(t) for t in self.acts[i])
            s1, s2, s3 = b.f1.get()[0], b.f2.get()[0], b.f3.get()[0]
            tagp = '%s.%d' % (self.name, i)
            n, h, w, _ = x_in.shape
            oh, ow = a2.shape[1], a2.shape[2]
            g_a2 = self.bufs.get('g_a2_%d_%d' % (oh, b.planes), a2.shape)
            g_a1 = self.bufs.get('g_a1_%d_%d' % (h, b.planes), a1.shape)
            # conv3 (1x1): g_a2 = [a2>0] * W3^T (g * s3)
            a2b = self.a2_bits[i] if getattr(self, 'a2_bits', None) else None
            mk3 = {'mask_bits': self._b(a2b)} if a2b is not None else {'mask': a2}
            ls, _ = ops.conv_dgrad(g, b.w3, g_a2, 1, 1, 0, **mk3, **fold(s3))
            self.bwd += [(l, tagp + '.conv3.dgrad') for l in ls]
            # conv2 (3x3, stride s): g_a1 = [a1>0] * dgrad(g_a2 * s2)
            if b.wino is not None:
                tile = wino_tile_for(b.wino.tile, n, h, w)
                v, m = self._wino_scratch(n, h, w, b.planes, b.planes, tile)
                self.bwd += ops.WinoConv(g_a2, b.wino.get(True, tile), g_a1, v, m, pro_scale=s2,
                                         mask=a1).launches(tagp + '.conv2.dgrad')
            else:
                ls, _ = ops.conv_dgrad(g_a2, b.w2, g_a1, 3, b.stride, 1, mask=a1, **fold(s2))
                self.bwd += [(l, tagp + '.conv2.dgrad') for l in ls]
            # conv1 (1x1) + identity / downsample fan-in, masked by the previous block's ReLU
            if i > 0:
                g_prev = self.bufs.get('g_blk_%d' % ((i - 1) & 1), x_in.shape)
                tgt, tmask, tres2 = g_prev, x_in, block_grads.get(i - 1)      # a loss term on block i-1's output
                tbits = self.out_bits[i - 1]
                tbits = None if tbits is None else self._b(tbits)
            else:
                tgt, tmask, tres2, tbits = dst, dst_mask, res2, dst_mask_bits
            # [x_in > 0] as nibbles where the forward wrote them (1/16 of the bytes), else the fp32 activation itself
            mk = {'mask_bits': tbits} if tbits is not None else {'mask': tmask}
            if b.has_ds:
                ls, _ = ops.conv_dgrad(g_a1, b.w1, tgt, 1, 1, 0, res2=tres2, **mk, **fold(s1))
                self.bwd += [(l, tagp + '.conv1.dgrad') for l in ls]
                ls, _ = ops.conv_dgrad(g, b.wd, tgt, 1, b.stride, 0, accumulate=True, **mk,
                                       **fold(b.fd.get()[0]))
                self.bwd += [(l, tagp + '.downsample.dgrad') for l in ls]
                flops += 2 * n * oh * ow * b.planes * 4 * b.cin
            else:
                ls, _ = ops.conv_dgrad(g_a1, b.w1, tgt, 1, 1, 0, res1=g, res2=tres2, **mk, **fold(s1))
                self.bwd += [(l, tagp + '.conv1.dgrad') for l in ls]
            flops += 2 * n * (h * w * b.planes * b.cin + oh * ow * b.planes * b.planes * 9
                              + oh * ow * b.planes * 4 * b.planes)
            g = tgt
        self.flops_bwd = flops


# =========================================================================================== student head
class _HeadConv(object):
    __slots__ = ('conv', 'bn', 'pad', 'relu', 'cin', 'cout', 'cs_in', 'cs_out', 'wc', 'wino', 'wino_f', 'wino_d', 'wino_w')


class HeadEngine(object):
    """Bottleneck4LargeResNet: eight bias-free 2x2 convs, each followed by a BatchNorm2d (+ReLU on four of them).
    Conv i reads the RAW output of conv i-1 and applies BN(+ReLU) i-1 on load; its epilogue emits the batch
    statistics of its own raw output (train mode) for hnd_bn_finalize."""

    def __init__(self, convs_bns):
        """convs_bns: list of (conv module, pad, following bn module, relu_after_bn)."""
        self.layers = []
        for conv, pad, bn, relu in convs_bns:
            hc = _HeadConv()
            hc.conv, hc.bn, hc.pad, hc.relu = conv, bn, pad, relu
            hc.cout, hc.cin = conv.weight.shape[0], conv.weight.shape[1]
            hc.cs_in, hc.cs_out = ops.chan_pad_of(hc.cin), ops.chan_pad_of(hc.cout)
            hc.wc = WeightCache(conv.weight)
            # Winograd F(6x6,2x2) / F(4x4,2x2) per DIRECTION (tools/bench_wino2.py, profiles/r03_bench_wino2.txt, batch 16):
            #   * both sides >= 128 channels (decoder 128->256, 256->256): forward x1.4-1.7, data gradient x1.4-1.7, weight
            #     gradient in the Winograd domain x1.5-2.0 -- everything;
            #   * 256 -> 64 (encoder conv2): forward x1.23 and, reusing its V, the weight gradient x2.1; its data gradient
            #     (GEMM depth 64) gains nothing -> forward + weight gradient only;
            #   * 64 -> 256 (encoder conv1): only the data gradient (GEMM depth 256) gains, x1.29;
            #   * 64 channels on both sides / 64 -> 128: the component GEMMs are HBM-bound and lose.
            deep = min(hc.cin, hc.cout) >= 128
            ok = bool(WINOGRAD and WINOGRAD_HEAD and 512 % hc.cout == 0)
            hc.wino_f = ok and (deep or (WINOGRAD2_6 and hc.cin >= 256 and hc.cout == 64))
            hc.wino_d = ok and (deep or (WINOGRAD2_6 and hc.cin == 64 and hc.cout >= 256))
            hc.wino = Wino2Cache(conv.weight) if (hc.wino_f or hc.wino_d) else None
            #   * 64 -> 256 (encoder conv1), round 4: the WEIGHT gradient in the Winograd domain as well, on an input
            #     transform of its own made in the backward pass (the forward stays direct): 2.9x fewer multiplies than
            #     the direct weight gradient (1.31 -> ~0.6 ms), and with both consumers of dy on transforms the BatchNorm
            #     backward apply of this 256-channel tensor fuses away (ops.wino26_bnbwd_step)
            hc.wino_w = bool(ok and WINO_WGRAD_OWN_V and not hc.wino_f and hc.wino_d and hc.cin == 64 and hc.cout % 256 == 0)
            self.layers.append(hc)
        self.bufs = None
        self.plan_key = None
        self.bwd_key = None
        self.flops_fwd = self.flops_bwd = 0
        self.encoder_len = 4          # convs 0..3 form the encoder, 4..7 the decoder (resnet_layer.py:42-65)
        self.out_provider = None      # see FrozenLayerEngine.out_provider

    def out_shape(self, x):
        n, h, w, _ = x.shape
        for hc in self.layers:
            h, w = ops.conv_out_size(h, 2, 1, hc.pad), ops.conv_out_size(w, 2, 1, hc.pad)
        return (n, h, w, self.layers[-1].cs_out)

    def bwd_out(self):
        return self.out

    def bwd_out_bits(self):
        return getattr(self, 'out_bits', None)

    def forward(self, x, training, codec=None):
        if self.bufs is None:
            self.bufs = Buffers(x.device)
        ops.pack_batch_begin()              # the ~28 operand re-packs of a training step go out as one launch
        try:
            for hc in self.layers:
                if hc.wino is not None:         # (packs are made by the plan, for the tile its geometry picks)
                    hc.wino.refresh(force=training)
                if not hc.wino_f:
                    hc.wc.get(False, hc.cs_in)
                hc.wc.refresh(force=training)            # forward / transposed / wgrad-side packs, whichever were made
        finally:
            ops.pack_batch_end()
        ptrs = tuple(t.data_ptr() for hc in self.layers
                     for t in (hc.bn.weight, hc.bn.bias, hc.bn.running_mean, hc.bn.running_var))
        self._out_buf = self.out_provider(self.out_shape(x)) if self.out_provider is not None else None
        key = (x.data_ptr(), tuple(x.shape), training, ptrs, None if self._out_buf is None else self._out_buf.data_ptr())
        if key != self.plan_key:
            self._build_forward(x, training)
            self.plan_key = key
            self.bwd_key = None
        b = self.bufs
        for i, hc in enumerate(self.layers):
            if not training:
                # eval-mode BN: fold running statistics (eps 1e-5) into the next prologue
                bn = hc.bn
                ops.fbn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                             eps=BN_EPS, cs=hc.cs_out, out=(self.scale[i], self.shift[i]))
            for l, tag in self.convs[i]:
                _run(l, tag)
            if codec is not None and i == self.encoder_len - 1:
                # eval-time bottleneck transformer on z = encoder output (logical [N, bch, H, W]); the codec
                # writes the dequantised tensor back into the same buffer, which the decoder plan reads
                z = logical(self.y[i], hc.cout)
                z._hnd = self.y[i]
                z2, _ = codec(z, None)
                if getattr(z2, '_hnd', None) is not self.y[i]:
                    # a transformer that builds a new tensor (the JPEG pair): lay it back into the plan's buffer
                    if not isinstance(z2, torch.Tensor) or tuple(z2.shape) != tuple(z.shape):
                        raise RuntimeError('bottleneck transformer must return a tensor of the bottleneck\'s shape')
                    self.y[i][..., :hc.cout].copy_(z2.to(self.y[i].device).permute(0, 2, 3, 1))
            if training:
                bn = hc.bn
                m = self.count[i]
                ops.bn_finalize(self.stats[i], self.ntiles[i], hc.cout, hc.cs_out, m, bn.weight.detach(),
                                bn.bias.detach(), bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                BN_MOMENTUM, BN_EPS, self.scale[i], self.shift[i], self.mean[i], self.rstd[i])
        ops.affine_relu(self.y[-1], self.scale[-1], self.shift[-1], self.out, self.layers[-1].relu,
                        mask_out=self.out_bits)
        return self.out

    def _build_forward(self, x, training):
        n, h, w, c = x.shape
        assert c == self.layers[0].cs_in
        b = self.bufs
        self.x = x
        self.y, self.stats, self.scale, self.shift, self.mean, self.rstd = [], [], [], [], [], []
        self.convs, self.count, self.ntiles, self.wino_fwd = [], [], [], {}
        cur, cur_scale, cur_shift, cur_relu = x, None, None, False
        flops = 0
        for i, hc in enumerate(self.layers):
            oh, ow = ops.conv_out_size(h, 2, 1, hc.pad), ops.conv_out_size(w, 2, 1, hc.pad)
            y = b.get('y%d' % i, (n, oh, ow, hc.cs_out))
            m = n * oh * ow
            t2 = wino2_tile_for(oh, ow)
            nt = ops.Wino2Conv.stats_blocks(n, oh, ow, hc.cs_out, t2) if hc.wino_f else ops.stats_tiles(m)
            st = b.get('stats%d' % i, (nt, 2, hc.cs_out)) if training else None
            sc, sh = b.get('scale%d' % i, (hc.cs_out,)), b.get('shift%d' % i, (hc.cs_out,))
            mu, rs = b.get('mean%d' % i, (hc.cs_out,)), b.get('rstd%d' % i, (hc.cs_out,))
            if hc.wino_f:
                v, mm = self._wino_scratch(n, oh, ow, hc.cs_in, hc.cs_out, t2)
                if training:        # the transformed input is kept: the weight gradient reuses it
                    v = b.get('wino_keep_v%d' % i,
                              (ops.Wino2Conv.scratch_elems(n, oh, ow, hc.cs_in, hc.cs_out, t2)[0],))
                with ops.emulation_unless(i in BX3_HEAD_FWD):
                    wl = ops.Wino2Conv(cur, hc.wino.get(False, t2), y, v, mm, hc.pad, pro_scale=cur_scale,
                                       pro_shift=cur_shift, pro_relu=cur_relu, stats=st)
                self.wino_fwd[i] = wl
                self.convs.append(wl.launches('layer1.conv%d' % i))
            else:
                with ops.emulation_unless(i in BX3_HEAD_FWD):
                    self.convs.append([(ops.conv_forward(cur, hc.wc.get(False, hc.cs_in), y, 2, 1, hc.pad,
                                                         pro_scale=cur_scale, pro_shift=cur_shift, pro_relu=cur_relu,
                                                         stats=st), 'layer1.conv%d' % i)])
            self.ntiles.append(nt)
            flops += 2 * m * hc.cout * 4 * hc.cin
            self.y.append(y)
            self.stats.append(st)
            self.scale.append(sc)
            self.shift.append(sh)
            self.mean.append(mu)
            self.rstd.append(rs)
            self.count.append(m)
            cur, cur_scale, cur_shift, cur_relu = y, sc, sh, hc.relu
            h, w = oh, ow
        self.out = b.get('out', (n, h, w, self.layers[-1].cs_out)) if self._out_buf is None else self._out_buf
        assert tuple(self.out.shape) == (n, h, w, self.layers[-1].cs_out)
        # ReLU mask of the layer output as nibbles, for the data gradient that enters this layer (MASK_BITS)
        self.out_bits = b.get('out_bits', (n, h, w, self.layers[-1].cs_out // 4), torch.uint8) \
            if (training and MASK_BITS and self.layers[-1].relu and self.layers[-1].cs_out % 4 == 0) else None
        self.flops_fwd = flops

    def _wino_scratch(self, n, oh, ow, cin, cout, tile=4):
        nv, nm = ops.Wino2Conv.scratch_elems(n, oh, ow, cin, cout, tile)
        cur = getattr(self, '_wino_need', (0, 0))
        self._wino_need = (max(nv, cur[0]), max(nm, cur[1]))
        return self.bufs.get('wino_v', (self._wino_need[0],)), self.bufs.get('wino_m', (self._wino_need[1],))

    def _wino_slab_elems(self, fw, hc):
        """split-K workspace of the grouped Winograd wgrad launch (sized for the largest layer that uses it)"""
        n, h, w, c, oh, ow = fw.geom
        d = ops.WgradDesc()
        tiles = n * ((oh + fw.tile - 1) // fw.tile) * ((ow + fw.tile - 1) // fw.tile)
        d.n, d.h, d.w_, d.cin, d.cin_real, d.oh, d.ow, d.cout, d.ldy = 1, 1, tiles, c, hc.cin, 1, tiles, hc.cout, hc.cout
        d.kh, d.kw, d.stride, d.pad, d.groups = 1, 1, 1, 0, fw.ww.ncomp
        need = (ops.wgrad_workspace_of(d) + 3) // 4
        self._wino_slab_need = max(need, getattr(self, '_wino_slab_need', 0))
        return self._wino_slab_need

    def bottleneck(self):
        """(raw conv output, scale, shift, relu) of the encoder's last conv = the bottleneck tensor z."""
        return self.y[3]

    # ------------------------------------------------------------------ split deployment (inference only)
    def forward_part(self, x, part):
        """Head/tail split (reference src/models/mimic/split_rcnn.py:13-37,162-185), eval-mode BatchNorm:
        part 'encoder': stem output -> raw bottleneck z (what crosses the link);
        part 'decoder': z -> layer1 output.  Each part has its own buffers and prebuilt launches."""
        lo, hi = (0, self.encoder_len) if part == 'encoder' else (self.encoder_len, len(self.layers))
        if self.bufs is None:
            self.bufs = Buffers(x.device)
        if not hasattr(self, 'parts'):
            self.parts = {}
        for hc in self.layers[lo:hi]:
            if hc.wino is not None:
                hc.wino.refresh()
            if not hc.wino_f:
                hc.wc.get(False, hc.cs_in)
            hc.wc.refresh()
        b = self.bufs
        first = max(lo - 1, 0)               # the decoder's first conv applies decoder.0 (BN of layer lo-1) on load
        ptrs = tuple(t.data_ptr() for hc in self.layers[first:hi] for t in (hc.bn.running_mean, hc.bn.weight))
        key = (x.data_ptr(), tuple(x.shape), ptrs)
        plan = self.parts.get(part)
        if plan is None or plan['key'] != key:
            n, h, w, c = x.shape
            assert c == self.layers[lo].cs_in, (c, self.layers[lo].cs_in)
            plan = {'key': key, 'convs': [], 'y': [], 'fold': {}}
            for i in range(first, hi):
                hc = self.layers[i]
                plan['fold'][i] = (b.get('%s.scale%d' % (part, i), (hc.cs_out,)),
                                   b.get('%s.shift%d' % (part, i), (hc.cs_out,)))
            cur = x
            pro = (None, None, False) if lo == 0 else plan['fold'][lo - 1] + (self.layers[lo - 1].relu,)
            for i in range(lo, hi):
                hc = self.layers[i]
                oh, ow = ops.conv_out_size(h, 2, 1, hc.pad), ops.conv_out_size(w, 2, 1, hc.pad)
                y = b.get('%s.y%d' % (part, i), (n, oh, ow, hc.cs_out))
                tag = 'layer1.%s.conv%d' % (part, i)
                if hc.wino_f:                    # same arithmetic as the unsplit model
                    t2 = wino2_tile_for(oh, ow)
                    v, mm = self._wino_scratch(n, oh, ow, hc.cs_in, hc.cs_out, t2)
                    plan['convs'] += ops.Wino2Conv(cur, hc.wino.get(False, t2), y, v, mm, hc.pad, pro_scale=pro[0],
                                                   pro_shift=pro[1], pro_relu=pro[2]).launches(tag)
                else:
                    plan['convs'].append((ops.conv_forward(cur, hc.wc.get(False, hc.cs_in), y, 2, 1, hc.pad,
                                                           pro_scale=pro[0], pro_shift=pro[1], pro_relu=pro[2]), tag))
                plan['y'].append(y)
                cur, pro, h, w = y, plan['fold'][i] + (hc.relu,), oh, ow
            plan['out'] = b.get('%s.out' % part, cur.shape) if part == 'decoder' else cur
            self.parts[part] = plan
        for i, (sc, sh) in plan['fold'].items():
            bn = self.layers[i].bn
            ops.fbn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, eps=BN_EPS,
                         cs=self.layers[i].cs_out, out=(sc, sh))
        for l, tag in plan['convs']:
            _run(l, tag)
        if part == 'decoder':
            last = len(self.layers) - 1
            ops.affine_relu(plan['y'][-1], plan['fold'][last][0], plan['fold'][last][1], plan['out'],
                            self.layers[last].relu)
        return plan['out']

    def grad_out_buffer(self):
        self.g_out = self.bufs.get('g_out', self.out.shape)
        return self.g_out

    def enc_grad_buffer(self, top):
        """where the loss gradient of a term on the BOTTLENECK TENSOR (a hook on ``layer1.encoder``: the raw output of the
        encoder's last conv) is written.  top: no term sits above it -- the backward then STARTS there (the decoder carries
        no gradient) and the buffer is that conv's dy itself; else it is added to the dy arriving from the decoder."""
        zi = self.encoder_len - 1
        self._enc = (self.bufs.get('g%d' % zi if top else 'enc_loss_grad', self.y[zi].shape), bool(top))
        if top:
            self.g_out = None
        return self._enc[0]

    def backward(self, grad_dst, need_input_grad):
        """self.g_out = grad w.r.t. the layer output. grad_dst: dict param -> destination tensor (or missing).
        Returns the buffer holding the gradient w.r.t. the layer input (stem output) if need_input_grad."""
        enc = getattr(self, '_enc', None)
        self._enc = None                    # (set again by the next forward's loss, if it still has such a term)
        enc_top = enc is not None and enc[1]
        zi = self.encoder_len - 1
        if enc_top and self.g_out is None:
            self.g_out = self.bufs.get('g_out', self.out.shape)       # (never read: the plan below skips the decoder)
        key = (self.g_out.data_ptr(), need_input_grad, tuple(sorted((id(p), t.data_ptr()) for p, t in grad_dst.items())),
               None if enc is None else (enc[0].data_ptr(), enc_top))
        if key != self.bwd_key:
            self._build_backward(grad_dst, need_input_grad)
            self.bwd_key = key
        b = self.bufs
        side = main = None
        if self.bwd_side and not PROFILE['enabled']:
            if getattr(self, '_wgrad_stream', None) is None:
                self._wgrad_stream = torch.cuda.Stream(device=self.g_out.device)
            side, main = self._wgrad_stream, torch.cuda.current_stream()
        for i in range(len(self.layers) - 1, -1, -1):
            hc, st = self.layers[i], self.bsteps[i]
            g = st['g']
            if enc_top and i >= zi:
                # the gradient enters at the bottleneck tensor: the decoder (and the BatchNorm behind the bottleneck, which
                # is decoder.0) has none -- its parameters get exact zeros, as autograd leaves them untouched
                for t in ([st['dgamma'], st['dbeta']] + ([grad_dst[hc.conv.weight]] if (i > zi and hc.conv.weight in grad_dst) else [])):
                    ops.fill(t, 0.0)
                if i > zi:
                    continue
                # i == zi: g (= enc[0]) already holds dy of the encoder's last conv
                for l, tag in st['wgrad']:
                    _run(l, tag)
                for l, tag in st['dgrad']:
                    _run(l, tag)
                continue
            if st['folded'] is None:
                ops.bn_bwd_reduce(g, self.y[i], self.scale[i], self.shift[i], self.mean[i], self.rstd[i], hc.relu,
                                  st['part'])
                part, ntiles = st['part'], st['ntiles']
            else:                   # the sums came with g, out of the output transform of conv i+1's data gradient
                part, ntiles = st['folded']
            ops.bn_bwd_finalize(part, ntiles, hc.cout, hc.cs_out, self.count[i], hc.bn.weight.detach(),
                                self.mean[i], self.rstd[i], st['dgamma'], st['dbeta'], st['k123'])
            if st['fused'] is not None:      # dy is never materialised: g, x -> V (data gradient) and Z (weight gradient)
                _run(st['fused'], 'layer1.conv%d.bnbwd_transforms' % i)
            else:
                ops.bn_bwd_apply(g, self.y[i], self.scale[i], self.shift[i], st['k123'], hc.relu, g)   # in place -> dy
                if enc is not None and i == zi:          # + the gradient of the term on the bottleneck tensor
                    ops.add_inplace(g, enc[0])
            if side is not None and st['wgrad']:
                # dy (or Z) of this layer is complete on the main stream here; the weight gradient reads it, the kept
                # forward V and buffers of its own, and writes only dW: it runs beside the data-gradient chain
                st['ev'].record(main)
                side.wait_event(st['ev'])
                with torch.cuda.stream(side):
                    for l, tag in st['wgrad']:
                        _run(l, tag)
            else:
                for l, tag in st['wgrad']:
                    _run(l, tag)
            for l, tag in st['dgrad']:
                _run(l, tag)
        self._side_pending = side
        return self.g_in if need_input_grad else None

    def join_wgrad_stream(self):
        """the weight gradients enqueued beside the data-gradient chain are complete for whatever the caller's stream
        does next (called after the stem's backward, which they also run beside)"""
        side = getattr(self, '_side_pending', None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            self._side_pending = None

    def _build_backward(self, grad_dst, need_input_grad):
        b = self.bufs
        self.bsteps = [None] * len(self.layers)
        self.bwd_side = bool(self.g_out.is_cuda and wgrad_stream_on())   # (the plan below gives Z buffers of their own)
        flops = 0
        nl = len(self.layers)
        # gradient buffers: g[i] = grad w.r.t. BN_i output (then, in place, w.r.t. raw conv_i output)
        gbuf = [b.get('g%d' % i, self.y[i].shape) for i in range(nl - 1)] + [self.g_out]
        self.g_in = b.get('g_in', self.x.shape) if need_input_grad else None
        slab_bytes = 0
        for i, hc in enumerate(self.layers):
            src = self.x if i == 0 else self.y[i - 1]
            n, h, w, _ = src.shape
            oh, ow = self.y[i].shape[1], self.y[i].shape[2]
            slab_bytes = max(slab_bytes, ops.wgrad_workspace_bytes(n, h, w, hc.cs_in, oh, ow, hc.cout, 2, 1, hc.pad))
        slabs = b.get('slabs', ((slab_bytes + 3) // 4,))
        for i, hc in enumerate(self.layers):
            st = {'folded': None, 'ev': torch.cuda.Event() if self.bwd_side else None}
            npix = self.count[i]
            st['g'] = gbuf[i]
            st['ntiles'] = ops.bn_bwd_ntiles(npix)
            st['part'] = b.get('bpart%d' % i, (st['ntiles'], 2, hc.cs_out))
            st['k123'] = b.get('k123_%d' % i, (3, hc.cs_out))
            st['dgamma'] = grad_dst.get(hc.bn.weight, None)
            st['dbeta'] = grad_dst.get(hc.bn.bias, None)
            if st['dgamma'] is None:
                st['dgamma'] = b.get('dgamma_scratch%d' % i, (hc.cout,))
            if st['dbeta'] is None:
                st['dbeta'] = b.get('dbeta_scratch%d' % i, (hc.cout,))
            src = self.x if i == 0 else self.y[i - 1]
            pro = (None, None, False) if i == 0 else (self.scale[i - 1], self.shift[i - 1], self.layers[i - 1].relu)
            dw = grad_dst.get(hc.conv.weight, None)
            st['wgrad'] = []
            wg_obj = dg_obj = None
            if dw is not None and i in self.wino_fwd and hc.cs_out == hc.cout and hc.cs_in == hc.cin:
                # Winograd-domain weight gradient: forward V x transformed dy, 25 grouped split-K reductions
                fw = self.wino_fwd[i]
                _, zbuf = self._wino_scratch(fw.geom[0], fw.geom[4], fw.geom[5], hc.cs_in, hc.cs_out, fw.tile)
                if self.bwd_side:       # Z must outlive the data gradient's M (which the shared scratch aliases)
                    zbuf = b.get('wino_z%d' % i, (ops.Wino2Conv.scratch_elems(fw.geom[0], fw.geom[4], fw.geom[5], hc.cs_in,
                                                                             hc.cs_out, fw.tile)[1],))
                sbuf = b.get('wino_s%d' % i, (fw.ww.ncomp * hc.cout * hc.cin,))
                wg_obj = ops.Wino2Wgrad(fw, gbuf[i], dw, zbuf, sbuf, b.get('wino_slabs', (self._wino_slab_elems(fw, hc),)))
                st['wgrad'] = wg_obj.launches('layer1.conv%d.wgrad' % i)
                flops += 2 * npix * hc.cout * 4 * hc.cin
            elif (dw is not None and hc.wino_w and hc.cs_out == hc.cout and hc.cs_in == hc.cin
                  and wino2_tile_for(self.y[i].shape[1], self.y[i].shape[2]) == 6):
                # Winograd-domain weight gradient on an input transform of its own (made here, in the backward pass)
                n_, oh_, ow_ = self.y[i].shape[0], self.y[i].shape[1], self.y[i].shape[2]
                vw = b.get('wino_vw%d' % i, (ops.Wino2InputTransform.scratch_elems(n_, oh_, ow_, hc.cs_in, 6),))
                own = ops.Wino2InputTransform(src, vw, hc.pad, hc.cout, 6, pro_scale=pro[0], pro_shift=pro[1],
                                              pro_relu=pro[2])
                _, zbuf = self._wino_scratch(n_, oh_, ow_, hc.cs_in, hc.cs_out, 6)
                if self.bwd_side:
                    zbuf = b.get('wino_z%d' % i, (ops.Wino2Conv.scratch_elems(n_, oh_, ow_, hc.cs_in, hc.cs_out, 6)[1],))
                sbuf = b.get('wino_s%d' % i, (own.ww.ncomp * hc.cout * hc.cin,))
                wg_obj = ops.Wino2Wgrad(own, gbuf[i], dw, zbuf, sbuf, b.get('wino_slabs', (self._wino_slab_elems(own, hc),)))
                wg = wg_obj.launches('layer1.conv%d.wgrad' % i)
                st['wgrad'] = [wg[0], own.step('layer1.conv%d.wgrad' % i)] + wg[1:]      # dy transform first (see `fused`)
                flops += 2 * npix * hc.cout * 4 * hc.cin
            elif dw is not None:
                st['wgrad'] = [(ops.conv_wgrad(src, gbuf[i], dw, 2, 1, hc.pad, pro_scale=pro[0], pro_shift=pro[1],
                                               pro_relu=pro[2], slabs=slabs), 'layer1.conv%d.wgrad' % i)]
                flops += 2 * npix * hc.cout * 4 * hc.cin
            st['dgrad'] = []
            tgt = gbuf[i - 1] if i > 0 else self.g_in
            if tgt is not None and hc.wino_d:
                nd, hd, wd, _ = tgt.shape
                t2 = wino2_tile_for(hd, wd)
                v, mm = self._wino_scratch(nd, hd, wd, hc.cs_out, hc.cs_in, t2)
                bwd_stats = None
                prev = self.layers[i - 1] if i > 0 else None
                if (FOLD_BNBWD_REDUCE and prev is not None and t2 == 6 and prev.cs_out == prev.cout
                        and 512 % prev.cs_out == 0 and tgt.shape[3] == prev.cs_out):
                    # tgt = g of layer i-1: its BatchNorm-backward sums come out of this launch's output transform
                    nblk = ops.Wino2Conv.stats_blocks(nd, hd, wd, prev.cs_out, 6)
                    part = b.get('bpart_folded%d' % (i - 1), (nblk, 2, prev.cs_out))
                    bwd_stats = (self.y[i - 1], self.scale[i - 1], self.shift[i - 1], self.mean[i - 1],
                                 self.rstd[i - 1], prev.relu, part)
                    self.bsteps[i - 1]['folded'] = (part, nblk)
                dg_obj = ops.Wino2Conv(gbuf[i], hc.wino.get(True, t2), tgt, v, mm, 1 - hc.pad, bwd_stats=bwd_stats)
                st['dgrad'] = dg_obj.launches('layer1.conv%d.dgrad' % i)
                flops += 2 * npix * hc.cout * 4 * hc.cin
            elif tgt is not None:
                pk = hc.wc.get(True, hc.cs_out)
                ls, _ = _dgrad_with_pack(gbuf[i], hc, tgt, pk)
                prev = self.layers[i - 1] if i > 0 else None
                if (FOLD_BNBWD_REDUCE and prev is not None and prev.cs_out == prev.cout and tgt.shape[3] == prev.cs_out
                        and ls[0].variant in _STATS_KERNELS):
                    # a tiled (or B-streamed emulated) launch: the BatchNorm-backward sums of layer i-1 come out of its
                    # epilogue (per 128 pixels)
                    nblk = ops.stats_tiles(tgt.shape[0] * tgt.shape[1] * tgt.shape[2])
                    part = b.get('bpart_folded%d' % (i - 1), (nblk, 2, prev.cs_out))
                    ls, _ = _dgrad_with_pack(gbuf[i], hc, tgt, pk, stats=part, bwd_stats=(
                        self.y[i - 1], self.scale[i - 1], self.shift[i - 1], self.mean[i - 1], self.rstd[i - 1], prev.relu))
                    assert ls[0].variant in _STATS_KERNELS, ls[0].variant
                    self.bsteps[i - 1]['folded'] = (part, nblk)
                st['dgrad'] = [(l, 'layer1.conv%d.dgrad' % i) for l in ls]
                flops += 2 * npix * hc.cout * 4 * hc.cin
            # both consumers of dy are F(6x6,2x2) transforms: fuse the BN-backward apply into them (conv6, conv7 at full size)
            st['fused'] = None
            if FUSE_BNBWD and wg_obj is not None and dg_obj is not None and wg_obj.tile == 6 and dg_obj.tile == 6:
                st['fused'] = ops.wino26_bnbwd_step(gbuf[i], self.y[i], self.scale[i], self.shift[i], st['k123'], hc.relu,
                                                    dg_obj, wg_obj)
                st['wgrad'], st['dgrad'] = st['wgrad'][1:], st['dgrad'][1:]       # (their own transforms are dropped)
            self.bsteps[i] = st
        self.flops_bwd = flops


# kernels whose epilogue can make BatchNorm statistics / backward sums per 128-pixel tile (hnd_conv_desc.stats, .bwd_x)
_STATS_KERNELS = ('igemm_128x128', 'igemm_128x64', 'bxs_128', 'bxs_64')


def _dgrad_with_pack(dy, hc, dx, pk, **kw):
    """stride-1 dgrad of a head conv using the cached transposed pack (re-packed every training step)."""
    n, h, w, ldc = dx.shape
    l = ops.conv_desc(dy, pk, dx, kh=2, kw=2, oh=h, ow=w, sh=1, dh=-1, bh=hc.pad, sw=1, dw=-1, bw=hc.pad, cout=ldc, **kw)
    return [l], [pk]


# =========================================================================================== FPN
class FpnEngine(object):
    def __init__(self, inner_blocks, layer_blocks):
        self.inner = [(m, WeightCache(m.weight)) for m in inner_blocks]
        self.layer = [(m, WeightCache(m.weight)) for m in layer_blocks]
        self.wino = [WinoCache(m.weight, use_winograd(m.weight.shape[1], m.weight.shape[0], 1))
                     if use_winograd(m.weight.shape[1], m.weight.shape[0], 1) else None for m in layer_blocks]
        self.bufs = None
        self.plan_key = None
        self.flops_fwd = 0

    def prepare(self, feats):
        """refresh the operands and (re)build the plan for these inputs without launching; returns the output list"""
        if self.bufs is None:
            self.bufs = Buffers(feats[0].device)
        for i, (m, wc) in enumerate(self.inner + self.layer):
            if i >= len(self.inner) and self.wino[i - len(self.inner)] is not None:
                self.wino[i - len(self.inner)].refresh()      # (packs are made by the plan, per level geometry)
                continue
            wc.get()
            wc.refresh()
        key = tuple((f.data_ptr(), tuple(f.shape)) for f in feats) + \
            tuple(m.bias.data_ptr() for m, _ in self.inner + self.layer)
        if key != self.plan_key:
            self.fwd = []
            nlev = len(feats)
            inner = [None] * nlev
            self.results = [None] * nlev
            flops = 0
            # Winograd scratch sized for the finest level up front (the loop runs coarse -> fine)
            n0, h0, w0, _ = feats[0].shape
            oc = self.layer[0][0].weight.shape[0]
            need = (ops.WinoConv.scratch_elems(n0, h0, w0, oc, oc, wino_tile_for(WINOGRAD, n0, h0, w0))
                    if any(w is not None for w in self.wino) else (0, 0))
            for i in range(nlev - 1, -1, -1):
                f = feats[i]
                n, h, w, c = f.shape
                m, wc = self.inner[i]
                inner[i] = self.bufs.get('inner%d' % i, (n, h, w, m.weight.shape[0]))
                up = inner[i + 1] if i + 1 < nlev else None
                self.fwd.append((ops.conv_forward(f, wc.get(), inner[i], 1, 1, 0, epi_shift=m.bias.detach(),
                                                  res1=up, res1_up=up is not None), 'fpn.inner%d' % i))
                ml, wl = self.layer[i]
                self.results[i] = self.bufs.get('p%d' % i, (n, h, w, ml.weight.shape[0]))
                if self.wino[i] is not None:            # 3x3 256->256 output conv: Winograd F(2x2,3x3)
                    tile = wino_tile_for(self.wino[i].tile, n, h, w)
                    nv, nm = ops.WinoConv.scratch_elems(n, h, w, ml.weight.shape[1], ml.weight.shape[0], tile)
                    need = (max(nv, need[0]), max(nm, need[1]))
                    v, mm = self.bufs.get('wino_v', (need[0],)), self.bufs.get('wino_m', (need[1],))
                    self.fwd += ops.WinoConv(inner[i], self.wino[i].get(False, tile), self.results[i], v, mm,
                                             epi_shift=ml.bias.detach()).launches('fpn.layer%d' % i)
                else:
                    self.fwd.append((ops.conv_forward(inner[i], wl.get(), self.results[i], 3, 1, 1,
                                                      epi_shift=ml.bias.detach()), 'fpn.layer%d' % i))
                flops += 2 * n * h * w * 256 * (c + 256 * 9)
            last = self.results[-1]
            n, h, w, c = last.shape
            self.pool = self.bufs.get('pool', (n, (h + 1) // 2, (w + 1) // 2, c))
            self.flops_fwd = flops
            self.plan_key = key
        return self.results + [self.pool]

    def term_grad_buffer(self, level):
        """where the loss gradient of a term on pyramid map `level` (a hook on ``backbone.fpn.layer_blocks.level``) is written"""
        return self.bufs.get('g_p%d' % level, self.results[level].shape)

    def backward(self, term_grads, sinks):
        """Gradient of loss terms on pyramid maps down to the layer outputs (the pyramid's own weights are frozen in every
        config: no weight gradients).  torchvision 0.4.2 FeaturePyramidNetwork.forward, backwards:
            P_i = conv3x3_i(inner_i),   inner_i = lateral_i(C_i) + nearest_up(inner_{i+1})
        term_grads: {level: dL/dP_level};  sinks: {level: (dst, mask, accumulate)} -- dL/dC_level is written (accumulate:
        added) to dst, masked by [mask > 0] when a mask tensor is given (the top layer's gradient buffer holds MASKED
        gradients).  Levels below the finest term level receive nothing."""
        assert term_grads and all(0 <= k < len(self.results) for k in term_grads)
        key = (tuple(sorted((k, t.data_ptr()) for k, t in term_grads.items())),
               tuple(sorted((k, v[0].data_ptr(), None if v[1] is None else v[1].data_ptr(), bool(v[2]))
                            for k, v in sinks.items())), self.plan_key)
        if key != getattr(self, 'bwd_key', None):
            self.bwd, kmin, nlev = [], min(term_grads), len(self.results)
            prev = None
            for i in range(kmin, nlev):
                n, h, w, c = self.results[i].shape
                g_inner = self.bufs.get('g_inner%d' % i, (n, h, w, c))
                have = False
                if i in term_grads:
                    ml, wl = self.layer[i]
                    if self.wino[i] is not None:
                        tile = wino_tile_for(self.wino[i].tile, n, h, w)
                        nv, nm = ops.WinoConv.scratch_elems(n, h, w, c, c, tile)
                        v, mm = self.bufs.get('wino_v_bwd', (nv,)), self.bufs.get('wino_m_bwd', (nm,))
                        self.bwd += ops.WinoConv(term_grads[i], self.wino[i].get(True, tile), g_inner, v,
                                                 mm).launches('fpn.layer%d.dgrad' % i)
                    else:
                        ls, _ = ops.conv_dgrad(term_grads[i], wl, g_inner, 3, 1, 1)
                        self.bwd += [(l, 'fpn.layer%d.dgrad' % i) for l in ls]
                    have = True
                if prev is not None:
                    self.bwd.append((ops._Step(lambda s, a=prev, b_=g_inner, acc=have: ops.upsample_nearest_bwd(a, b_, acc),
                                               'upsample_nearest_bwd', 4 * (prev.numel() + g_inner.numel())),
                                     'fpn.inner%d.topdown.bwd' % (i - 1)))
                dst, mask, accumulate = sinks[i]
                kw = {'mask': mask} if mask is not None else {}
                ls, _ = ops.conv_dgrad(g_inner, self.inner[i][1], dst, 1, 1, 0, accumulate=bool(accumulate), **kw)
                self.bwd += [(l, 'fpn.inner%d.dgrad' % i) for l in ls]
                prev = g_inner
            self.bwd_key = key
        for l, tag in self.bwd:
            _run(l, tag)

    def forward(self, feats):
        """feats: list of NHWC layer outputs (fine -> coarse).  Returns list of NHWC pyramid maps + 'pool'."""
        outs = self.prepare(feats)
        for l, tag in self.fwd:
            _run(l, tag)
        ops.subsample2(self.results[-1], self.pool)
        return outs


# =========================================================================================== shared trunk
# Set by DistillationBox around its teacher + student calls: the SharedTrunk both backbones then run layers 2-4 (+ the
# feature pyramid) through, or None.
MERGE = {'trunk': None}
# Off by default -- measured, round 4 (profiles/r04_trunk_sweep.txt, same box, batch 16): the shared pass takes 1.0-1.5 ms
# of kernel time off the step (437 -> ~350 launches, layer3 / layer4 on the persistent kernels) but the two networks no
# longer run as two independent chains on two streams, and that overlap is worth ~3 ms: 94.5-95.3 ms with separate passes,
# 96.0-97.1 ms merged (from layer2, layer3 or layer4 alike).  HND_MERGE_TRUNK=1 turns it on; results are bit-identical.
MERGE_TRUNK = os.environ.get('HND_MERGE_TRUNK', '0') != '0'
# first layer of the shared pass: earlier layers run per network (teacher || student on two streams, which is where the
# HBM-bound high-resolution kernels find an MFMA-bound partner), later ones over the concatenated batch
MERGE_FROM = os.environ.get('HND_MERGE_FROM', 'layer3')


class SharedTrunk(object):
    """Layers 2-4 and the feature pyramid of teacher AND student as ONE pass over the concatenated batch.

    The student's layer2-4 / FPN are copies of the teacher's (reference src/models/org/rcnn.py:444-450 loads the
    teacher's COCO weights into the student with strict=False; yaml ``frozen_modules`` keeps them frozen), so the two
    networks used to launch every one of those convs twice with the same operand at half the GEMM height.  Here
    layer1 of the teacher and layer1 (the bottleneck head) of the student write straight into the two halves of one
    ``[2N, H, W, 256]`` buffer (``out_provider``) -- and so do their layer2 passes when the shared pass starts at layer3
    (``HND_MERGE_FROM``, the default: see MERGE_FROM) -- and one FrozenLayerEngine per shared layer / one FpnEngine runs
    over all 2N images: half the launches, twice the rows per launch -- which is what lets the persistent GEMM kernels take the
    layer3 / layer4 launches that were too short for them at N images.  No image-to-image coupling exists in these
    layers (FrozenBatchNorm), the Winograd tile depends on the map size only and every GEMM variant is bit-identical on
    the same descriptor, so both halves hold exactly the bits the separate passes produced (tests).

    Protocol inside one DistillationBox.forward (``MERGE['trunk']`` set): the TEACHER is called first.  Its stem and
    layer1 are launched (on the teacher stream); its layer2-4 / FPN module calls only build the plans and return VIEWS
    of the teacher half of the trunk's output buffers -- forward hooks fire and stash them, their contents arrive with
    the student's call.  The STUDENT's stem and head are launched on the main stream; at its layer2 call the main
    stream waits for the teacher stream and the merged layer is launched, and so on; the student's module calls return
    the student half.  The backward plan runs over the student half only (``bslice``).

    Used only while both weight sets are bit-equal (checked on the device whenever a tensor's version or address
    changed) and frozen; otherwise DistillationBox falls back to two separate passes."""
    ALL = ('layer1', 'layer2', 'layer3', 'layer4')

    def __init__(self, t_backbone, s_backbone, first=None):
        first = first or MERGE_FROM
        if first not in self.ALL[1:]:
            raise ValueError('HND_MERGE_FROM must be one of layer2 / layer3 / layer4, got %r' % (first,))
        self.LAYERS = self.ALL[self.ALL.index(first):]          # the layers of the shared pass
        self.FRONT = self.ALL[:self.ALL.index(first)]           # per-network layers that write into shared buffers
        self.backbones = (t_backbone, s_backbone)
        self.bodies = (t_backbone.body, s_backbone.body)
        s_body, s_fpn = s_backbone.body, s_backbone.fpn
        self.engines = OrderedDict((name, FrozenLayerEngine(list(s_body[name]), name)) for name in self.LAYERS)
        self.fpn_engine = FpnEngine(list(s_fpn.inner_blocks), list(s_fpn.layer_blocks))
        self.bufs = None
        self.front = {}                 # front layer -> [2N, H, W, C]: its outputs, teacher half first
        self.n = None
        self.t_stream = None
        self.seen = [False, False]      # which role has delivered its layer1 output in the current scope
        self._equal_key, self._equal = None, False

    # ---- eligibility
    @staticmethod
    def structure_ok(t_backbone, s_backbone):
        from . import hipnn
        try:
            tb, sb = t_backbone.body, s_backbone.body
            for name in SharedTrunk.ALL[1:]:
                if not (isinstance(tb[name], hipnn.ResLayer) and isinstance(sb[name], hipnn.ResLayer)
                        and len(tb[name]) == len(sb[name])):
                    return False
                if any(p.requires_grad for m in (tb[name], sb[name]) for p in m.parameters()):
                    return False
            if not (isinstance(tb['layer1'], hipnn.ResLayer) and hasattr(sb['layer1'], 'head_engine')
                    and not getattr(sb['layer1'], 'uses_ext_encoder', False)):
                return False
            if any(p.requires_grad for m in (t_backbone.fpn, s_backbone.fpn) for p in m.parameters()):
                return False
            return t_backbone.run_fpn == s_backbone.run_fpn
        except (KeyError, AttributeError, TypeError):
            return False

    def _tensor_pairs(self):
        (tb, sb), (tf, sf) = self.bodies, (self.backbones[0].fpn, self.backbones[1].fpn)
        pairs = []
        for tm, sm in [(tb[n], sb[n]) for n in self.ALL[1:]] + [(tf, sf)]:
            t_sd, s_sd = tm.state_dict(keep_vars=True), sm.state_dict(keep_vars=True)
            if list(t_sd) != list(s_sd):
                return None
            pairs += [(t_sd[k], s_sd[k]) for k in t_sd]
        return pairs

    RECHECK_EVERY = 256          # steps between unconditional device re-checks of the cached verdict

    def invalidate(self):
        """forget the cached 'equal' verdict: call after editing frozen weights through ``.data`` (which moves neither
        a version counter nor an address)"""
        self._equal_key = None

    def weights_equal(self):
        """every tensor of layer2-4 and the FPN bit-equal between the two networks (re-checked on the device when a
        version counter or an address moved -- load_state_dict, .to(), an in-place edit -- and every RECHECK_EVERY
        steps regardless, so an edit through ``.data`` cannot go unnoticed for long; invalidate() forces it)"""
        pairs = self._tensor_pairs()
        if pairs is None:
            return False
        self._checks = getattr(self, '_checks', 0) + 1
        if self._checks % self.RECHECK_EVERY == 0:
            self._equal_key = None
        key = tuple((a.data_ptr(), a._version, b.data_ptr(), b._version) for a, b in pairs)
        if key != self._equal_key:
            self._equal = all(a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b) for a, b in pairs)
            self._equal_key = key
        return self._equal

    # ---- one step
    def begin(self):
        self.seen = [False, False]

    def role_of(self, body):
        return 0 if body is self.bodies[0] else (1 if body is self.bodies[1] else None)

    def slot_provider(self, role, name):
        """out_provider of front layer `name` of network `role`: its half of the shared [2N, ...] buffer"""
        def provide(shape):
            n = int(shape[0])
            full = (2 * n,) + tuple(int(v) for v in shape[1:])
            if self.bufs is None:
                self.bufs = Buffers(torch.device('cuda', torch.cuda.current_device()))
            cur = self.front.get(name)
            if role == 1 and cur is not None and tuple(cur.shape) != full:
                raise RuntimeError('SharedTrunk: teacher and student %s outputs differ in shape (%s vs %s)'
                                   % (name, tuple(cur.shape), full))
            self.front[name] = self.bufs.get('front_' + name, full)
            self.n = n
            return self.front[name][role * n:(role + 1) * n]
        return provide

    def delivered(self, role):
        """the front layers of `role` have been enqueued into their halves of the shared buffers (current stream)"""
        if role == 0:
            self.t_stream = torch.cuda.current_stream()
        elif not self.seen[0]:
            raise RuntimeError('SharedTrunk: the student reached the shared layers before the teacher delivered its '
                               'front layers (DistillationBox calls the teacher first)')
        self.seen[role] = True

    def _half(self, t, role):
        return t[role * self.n:(role + 1) * self.n]

    def layer_forward(self, name, role):
        eng = self.engines[name]
        idx = self.LAYERS.index(name)
        x = self.front[self.FRONT[-1]] if idx == 0 else self.engines[self.LAYERS[idx - 1]].out
        # the same answer in both roles (the teacher's views must survive until the student's launch): keep every block's
        # output when the student trains or when a Bottleneck of either network carries a forward hook
        keep = self.bodies[1].needs_backward() or any(bool(b[name].hooked_blocks()) for b in self.bodies)
        eng.bslice = (self.n, 2 * self.n)
        if role == 0:
            return self._half(eng.prepare(x, keep), 0)         # a view; filled by the student's call
        if idx == 0:
            cur = torch.cuda.current_stream()
            if self.t_stream is not None and self.t_stream != cur:
                cur.wait_stream(self.t_stream)
        return self._half(eng.forward(x, keep), 1)

    def fpn_forward(self, role):
        feats = [self.front[n] for n in self.FRONT] + [self.engines[n].out for n in self.LAYERS]
        if role == 0:
            return [self._half(t, 0) for t in self.fpn_engine.prepare(feats)]
        return [self._half(t, 1) for t in self.fpn_engine.forward(feats)]


# =========================================================================================== neural filter
class _FilterConv(object):
    __slots__ = ('conv', 'bn', 'k', 'stride', 'cin', 'cout', 'cs_in', 'cs_out', 'wc')


class FilterEngine(object):
    """Ext4ResNet (reference src/models/ext/classifier.py:16-37): AdaptiveAvgPool(64x64) -> three biased convs,
    each followed by BatchNorm + ReLU (applied on load by the next consumer) -> AdaptiveAvgPool(8x8) -> Linear.
    Convolutions, BatchNorm statistics and their backward reuse the distillation kernels; the bias rides in the
    conv epilogue (epi_shift) so the BN statistics emitted by the same epilogue include it."""

    def __init__(self, extractor, linear):
        mods = list(extractor)
        self.pool0, self.pool1 = tuple(mods[0].output_size), tuple(mods[-1].output_size)
        self.layers = []
        for i, m in enumerate(mods):
            if isinstance(m, torch.nn.Conv2d):
                fc = _FilterConv()
                fc.conv, fc.bn = m, mods[i + 1]
                assert isinstance(fc.bn, torch.nn.BatchNorm2d) and isinstance(mods[i + 2], torch.nn.ReLU)
                assert m.kernel_size[0] == m.kernel_size[1] and m.padding == (0, 0) and m.bias is not None
                fc.k, fc.stride = m.kernel_size[0], m.stride[0]
                fc.cout, fc.cin = m.weight.shape[0], m.weight.shape[1]
                fc.cs_in, fc.cs_out = ops.chan_pad_of(fc.cin), ops.chan_pad_of(fc.cout)
                fc.wc = WeightCache(m.weight)
                self.layers.append(fc)
        self.linear = linear
        self.bufs = None
        self.plan_key = None
        self.bwd_key = None

    def params(self):
        out = []
        for fc in self.layers:
            out += [fc.conv.weight, fc.conv.bias, fc.bn.weight, fc.bn.bias]
        return out + [self.linear.weight, self.linear.bias]

    # ------------------------------------------------------------------ forward
    def forward(self, x, training):
        """x: stem output, NHWC [N, H, W, 64].  Returns [N, 2] logits (training) or softmax probabilities."""
        if self.bufs is None:
            self.bufs = Buffers(x.device)
        for fc in self.layers:
            fc.wc.get(False, fc.cs_in)
            fc.wc.refresh(force=training)
        ptrs = tuple(t.data_ptr() for fc in self.layers
                     for t in (fc.conv.bias, fc.bn.weight, fc.bn.bias, fc.bn.running_mean, fc.bn.running_var))
        key = (x.data_ptr(), tuple(x.shape), training, ptrs)
        if key != self.plan_key:
            self._build_forward(x, training)
            self.plan_key = key
            self.bwd_key = None
        ops.adaptive_avgpool_fwd(x, self.p0)
        for i, fc in enumerate(self.layers):
            self.bias[i][:fc.cout].copy_(fc.conv.bias.detach())          # pad channels keep bias 0
            if not training:
                bn = fc.bn
                ops.fbn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, eps=BN_EPS,
                             cs=fc.cs_out, out=(self.scale[i], self.shift[i]))
            _run(self.convs[i], 'ext.conv%d' % i)
            if training:
                bn, m = fc.bn, self.count[i]
                ops.bn_finalize(self.stats[i], ops.stats_tiles(m), fc.cout, fc.cs_out, m, bn.weight.detach(),
                                bn.bias.detach(), bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                BN_MOMENTUM, BN_EPS, self.scale[i], self.shift[i], self.mean[i], self.rstd[i])
        ops.affine_relu(self.y[-1], self.scale[-1], self.shift[-1], self.a_last, True)
        ops.adaptive_avgpool_fwd(self.a_last, self.p1)
        ops.linear_fwd(self.p1, self.layers[-1].cout, self.linear.weight.detach(), self.linear.bias.detach(),
                       self.logits)
        if training:
            return self.logits
        ops.softmax_rows(self.logits, self.probs)
        return self.probs

    def _build_forward(self, x, training):
        n, h, w, c = x.shape
        assert c == self.layers[0].cs_in, (c, self.layers[0].cs_in)
        b = self.bufs
        self.x = x
        self.p0 = b.get('p0', (n, self.pool0[0], self.pool0[1], c))
        self.y, self.stats, self.scale, self.shift, self.mean, self.rstd = [], [], [], [], [], []
        self.convs, self.count, self.bias = [], [], []
        cur, cur_scale, cur_shift, cur_relu = self.p0, None, None, False
        h, w = self.pool0
        for i, fc in enumerate(self.layers):
            oh, ow = ops.conv_out_size(h, fc.k, fc.stride, 0), ops.conv_out_size(w, fc.k, fc.stride, 0)
            y = b.get('y%d' % i, (n, oh, ow, fc.cs_out))
            m = n * oh * ow
            st = b.get('stats%d' % i, (ops.stats_tiles(m), 2, fc.cs_out)) if training else None
            bias = b.get('bias%d' % i, (fc.cs_out,))
            bias.zero_()
            sc, sh = b.get('scale%d' % i, (fc.cs_out,)), b.get('shift%d' % i, (fc.cs_out,))
            mu, rs = b.get('mean%d' % i, (fc.cs_out,)), b.get('rstd%d' % i, (fc.cs_out,))
            self.convs.append(ops.conv_forward(cur, fc.wc.get(False, fc.cs_in), y, fc.k, fc.stride, 0,
                                               pro_scale=cur_scale, pro_shift=cur_shift, pro_relu=cur_relu,
                                               epi_shift=bias, stats=st))
            self.y.append(y)
            self.stats.append(st)
            self.bias.append(bias)
            self.scale.append(sc)
            self.shift.append(sh)
            self.mean.append(mu)
            self.rstd.append(rs)
            self.count.append(m)
            cur, cur_scale, cur_shift, cur_relu = y, sc, sh, True
            h, w = oh, ow
        last = self.layers[-1]
        self.a_last = b.get('a_last', (n, h, w, last.cs_out))
        self.p1 = b.get('p1', (n, self.pool1[0], self.pool1[1], last.cs_out))
        nout = self.linear.weight.shape[0]
        self.logits = b.get('logits', (n, nout))
        self.probs = b.get('probs', (n, nout))

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits, grad_dst):
        """dlogits [N, nout] (contiguous); grad_dst: {parameter: destination tensor}."""
        key = (dlogits.data_ptr(), tuple(sorted((id(p), t.data_ptr()) for p, t in grad_dst.items())))
        if key != self.bwd_key:
            self._build_backward(grad_dst)
            self.bwd_key = key
        last = self.layers[-1]
        ops.linear_bwd(self.p1, last.cout, self.linear.weight.detach(), dlogits, grad_dst[self.linear.weight],
                       grad_dst[self.linear.bias], self.dp1)
        ops.adaptive_avgpool_bwd(self.dp1, self.g[-1])
        for i in range(len(self.layers) - 1, -1, -1):
            fc, st, g = self.layers[i], self.bsteps[i], self.g[i]
            ops.bn_bwd_reduce(g, self.y[i], self.scale[i], self.shift[i], self.mean[i], self.rstd[i], True, st['part'])
            ops.bn_bwd_finalize(st['part'], st['ntiles'], fc.cout, fc.cs_out, self.count[i], fc.bn.weight.detach(),
                                self.mean[i], self.rstd[i], grad_dst[fc.bn.weight], grad_dst[fc.bn.bias], st['k123'])
            ops.bn_bwd_apply(g, self.y[i], self.scale[i], self.shift[i], st['k123'], True, g)     # in place -> dy
            ops.channel_sum(g, fc.cout, grad_dst[fc.conv.bias], self.sum_scratch)
            _run(st['wgrad'], 'ext.conv%d.wgrad' % i)
            for l in st['dgrad']:
                _run(l, 'ext.conv%d.dgrad' % i)

    def _build_backward(self, grad_dst):
        b = self.bufs
        missing = [p for p in self.params() if p not in grad_dst]
        if missing:
            raise RuntimeError('the neural filter trains all of its %d tensors together (ext_runner.py:196-197); '
                               '%d have requires_grad=False' % (len(self.params()), len(missing)))
        self.dp1 = b.get('dp1', self.p1.shape)
        self.sum_scratch = b.get('sum_scratch', (ops.channel_sum_scratch_elems(max(fc.cout for fc in self.layers)),))
        self.g = [b.get('g%d' % i, self.y[i].shape) for i in range(len(self.layers))]
        slab_bytes = 0
        for i, fc in enumerate(self.layers):
            src = self.p0 if i == 0 else self.y[i - 1]
            n, h, w, _ = src.shape
            slab_bytes = max(slab_bytes, ops.wgrad_workspace_bytes(n, h, w, fc.cs_in, self.y[i].shape[1],
                                                                   self.y[i].shape[2], fc.cout, fc.k, fc.stride, 0))
        slabs = b.get('slabs', ((slab_bytes + 3) // 4,))
        self.bsteps = []
        for i, fc in enumerate(self.layers):
            st = {'ntiles': ops.bn_bwd_ntiles(self.count[i])}
            st['part'] = b.get('bpart%d' % i, (st['ntiles'], 2, fc.cs_out))
            st['k123'] = b.get('k123_%d' % i, (3, fc.cs_out))
            src = self.p0 if i == 0 else self.y[i - 1]
            pro = (None, None, False) if i == 0 else (self.scale[i - 1], self.shift[i - 1], True)
            st['wgrad'] = ops.conv_wgrad(src, self.g[i], grad_dst[fc.conv.weight], fc.k, fc.stride, 0,
                                         pro_scale=pro[0], pro_shift=pro[1], pro_relu=pro[2], slabs=slabs)
            st['dgrad'] = []
            if i > 0:
                st['dgrad'], _ = ops.conv_dgrad(self.g[i], fc.wc, self.g[i - 1], fc.k, fc.stride, 0)
            self.bsteps.append(st)
