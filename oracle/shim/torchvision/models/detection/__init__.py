from oracle.tv042 import TVFasterRCNN as FasterRCNN, TVMaskRCNN as MaskRCNN, TVKeypointRCNN as KeypointRCNN  # noqa
from . import (backbone_utils, faster_rcnn, image_list, keypoint_rcnn, mask_rcnn, roi_heads, rpn,  # noqa
               transform, _utils)
