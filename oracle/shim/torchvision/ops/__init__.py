from oracle.tv042 import MultiScaleRoIAlign  # noqa
from . import misc, feature_pyramid_network  # noqa
