from oracle.tv042 import MultiScaleRoIAlign  # noqa
from . import misc, feature_pyramid_network  # noqa
from oracle.tv042_det import nms, batched_nms, roi_align, box_area, clip_boxes_to_image, remove_small_boxes  # noqa
from . import boxes  # noqa
