from oracle.tv042_det import nms, batched_nms, box_area, clip_boxes_to_image, remove_small_boxes  # noqa
