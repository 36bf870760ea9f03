from oracle.tv042 import GeneralizedRCNNTransform, resize_boxes, resize_keypoints  # noqa
