from . import resnet, utils, _utils, detection  # noqa
