"""COCO-format detection datasets without pycocotools (role of the reference's src/utils/coco_util.py).

The image has neither pycocotools nor torchvision, so the annotation file is indexed with the json module and
polygons are rasterised with PIL.  What the distillation / neural-filter steps consume is reproduced exactly --
decoded RGB image, ``boxes`` (xywh -> xyxy, clipped, degenerate ones dropped), ``labels``, ``keypoints``,
``image_id``, ``area``, ``iscrowd`` (reference :50-103) and the "has a valid annotation" filter (:106-144).
``masks`` are PIL polygon fills: pycocotools' rasteriser can differ on boundary pixels, and RLE segmentations of
non-crowd objects are not decoded (zero mask) -- masks are not read by any loss on this build's paths.
COCO evaluation (``convert_to_coco_api`` / ``CocoEvaluator``) stays out.
"""
import copy
import json
import os
from collections import defaultdict
from io import BytesIO

import numpy as np
import torch
import torch.utils.data
from PIL import Image, ImageDraw

from ..structure.transformer import Compose


class CocoIndex(object):
    """the slice of pycocotools.coco.COCO that detection datasets use"""

    def __init__(self, ann_file):
        with open(ann_file) as fp:
            data = json.load(fp)
        self.imgs = {im['id']: im for im in data.get('images', [])}
        self.cats = {c['id']: c for c in data.get('categories', [])}
        self.img_to_anns = defaultdict(list)
        for ann in data.get('annotations', []):
            self.img_to_anns[ann['image_id']].append(ann)

    def image_ids(self):
        return sorted(self.imgs)

    def annotations(self, image_id):
        return self.img_to_anns.get(image_id, [])


class FilterAndRemapCocoCategories(object):
    def __init__(self, categories, remap=True):
        self.categories, self.remap = categories, remap

    def __call__(self, image, target):
        anno = [obj for obj in target['annotations'] if obj['category_id'] in self.categories]
        if self.remap:
            anno = copy.deepcopy(anno)
            for obj in anno:
                obj['category_id'] = self.categories.index(obj['category_id'])
        target['annotations'] = anno
        return image, target


def convert_coco_poly_to_mask(segmentations, height, width):
    """reference :32-45: ``frPyObjects(polygons) -> decode -> any over the parts`` per object; rasterised by
    utils/mask_util.py (maskApi's polygon rule: a w x h rectangle covers exactly w*h pixels)"""
    from . import mask_util
    masks = []
    for polygons in segmentations:
        if isinstance(polygons, list):
            m = mask_util.polygons_to_mask(polygons, height, width)
        elif polygons:
            m = mask_util.segmentation_to_mask(polygons, height, width)
        else:
            m = np.zeros((height, width), dtype=bool)
        masks.append(torch.from_numpy(m.astype(np.uint8)))
    if masks:
        return torch.stack(masks, dim=0)
    return torch.zeros((0, height, width), dtype=torch.uint8)


class ConvertCocoPolysToMask(object):
    """raw COCO annotation list -> the tensors a detector takes (reference :50-103)"""

    def __call__(self, image, target):
        w, h = image.size
        anno = [obj for obj in target['annotations'] if obj.get('iscrowd', 0) == 0]
        boxes = torch.as_tensor([obj['bbox'] for obj in anno], dtype=torch.float32).reshape(-1, 4)
        boxes[:, 2:] += boxes[:, :2]
        boxes[:, 0::2].clamp_(min=0, max=w)
        boxes[:, 1::2].clamp_(min=0, max=h)
        classes = torch.tensor([obj['category_id'] for obj in anno], dtype=torch.int64)
        masks = convert_coco_poly_to_mask([obj.get('segmentation', []) for obj in anno], h, w)
        keypoints = None
        if anno and 'keypoints' in anno[0]:
            keypoints = torch.as_tensor([obj['keypoints'] for obj in anno], dtype=torch.float32)
            if keypoints.shape[0]:
                keypoints = keypoints.view(keypoints.shape[0], -1, 3)
        keep = (boxes[:, 3] > boxes[:, 1]) & (boxes[:, 2] > boxes[:, 0])
        out = {'boxes': boxes[keep], 'labels': classes[keep], 'masks': masks[keep],
               'image_id': torch.tensor([target['image_id']])}
        if keypoints is not None:
            out['keypoints'] = keypoints[keep]
        out['area'] = torch.tensor([obj['area'] for obj in anno])
        out['iscrowd'] = torch.tensor([obj.get('iscrowd', 0) for obj in anno])
        return image, out


def has_only_empty_bbox(anno):
    return all(any(o <= 1 for o in obj['bbox'][2:]) for obj in anno)


def count_visible_keypoints(anno):
    return sum(sum(1 for v in ann['keypoints'][2::3] if v > 0) for ann in anno)


def has_valid_annotation(anno, min_keypoints_per_image=10):
    if len(anno) == 0 or has_only_empty_bbox(anno):
        return False
    if 'keypoints' not in anno[0]:
        return True
    return count_visible_keypoints(anno) >= min_keypoints_per_image


class ExtCocoDetection(torch.utils.data.Dataset):
    """(PIL RGB image, annotations) -> transforms; optional JPEG re-compression of the input (``jpeg_quality``)."""

    def __init__(self, img_folder, ann_file, transforms, jpeg_quality=None):
        self.root, self.coco = img_folder, CocoIndex(ann_file)
        self.ids = self.coco.image_ids()
        self.additional_transforms = transforms
        self.jpeg_quality = jpeg_quality if jpeg_quality is not None and 1 <= jpeg_quality <= 95 else None

    def __len__(self):
        return len(self.ids)

    def get_height_and_width(self, index):
        info = self.coco.imgs[self.ids[index]]
        return info['height'], info['width']

    def __getitem__(self, index):
        img_id = self.ids[index]
        info = self.coco.imgs[img_id]
        img = Image.open(os.path.join(self.root, info['file_name'])).convert('RGB')
        if self.jpeg_quality is not None:
            buf = BytesIO()
            img.save(buf, 'JPEG', quality=self.jpeg_quality)
            img = Image.open(buf)
        target = {'image_id': img_id, 'annotations': self.coco.annotations(img_id)}
        if self.additional_transforms is not None:
            img, target = self.additional_transforms(img, target)
        return img, target


def remove_images_without_annotations(dataset, cat_list=None):
    keep = []
    for ds_idx, img_id in enumerate(dataset.ids):
        anno = dataset.coco.annotations(img_id)
        if cat_list:
            anno = [obj for obj in anno if obj['category_id'] in cat_list]
        if has_valid_annotation(anno):
            keep.append(ds_idx)
    return torch.utils.data.Subset(dataset, keep)


def get_coco(img_dir_path, ann_file_path, transforms, remove_non_annotated_imgs, jpeg_quality=None):
    chain = [ConvertCocoPolysToMask()] + ([transforms] if transforms is not None else [])
    dataset = ExtCocoDetection(os.path.expanduser(img_dir_path), os.path.expanduser(ann_file_path),
                               transforms=Compose(chain), jpeg_quality=jpeg_quality)
    if remove_non_annotated_imgs:
        dataset = remove_images_without_annotations(dataset)
    return dataset


def get_coco_api_from_dataset(dataset):
    """reference :198-206 lives in this module; the ground-truth index it returns is built by coco_eval_util"""
    from .coco_eval_util import get_coco_api_from_dataset as build
    return build(dataset)


def convert_to_coco_api(ds):
    """reference :148-195: a dataset of (image, target) pairs -> ground-truth index (same builder: a dataset without
    ``.coco`` annotations is walked target by target)"""
    return get_coco_api_from_dataset(ds)
