from oracle.tv042 import FrozenBatchNorm2d, interpolate, MiscConv2d as Conv2d, MiscConvTranspose2d as ConvTranspose2d  # noqa
