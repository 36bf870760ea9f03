from oracle.pycoco_r import COCO  # noqa
