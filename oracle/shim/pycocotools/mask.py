"""pycocotools.mask over the restatement in oracle/pycoco_r.py"""
from oracle.pycoco_r import mask as _m

iou, encode, decode, area, toBbox, frPyObjects, merge = (_m.iou, _m.encode, _m.decode, _m.area, _m.toBbox,
                                                          _m.frPyObjects, _m.merge)
