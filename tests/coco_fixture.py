"""Writes a tiny COCO-format dataset (PNG images + person_keypoints-style json) for loader tests."""
import json
import os

import numpy as np
from PIL import Image


def write_tiny_coco(root, sizes=((48, 64), (64, 48), (40, 80), (56, 56), (32, 96), (60, 62), (50, 70), (70, 50))):
    os.makedirs(os.path.join(root, 'images'), exist_ok=True)
    rng = np.random.RandomState(7)
    images, annotations = [], []
    ann_id = 1
    for i, (h, w) in enumerate(sizes):
        name = 'img_%02d.png' % i
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, 'images', name))
        images.append({'id': 100 + i, 'file_name': name, 'height': h, 'width': w})
        if i == 3:
            continue                                    # an image without annotations
        x, y, bw, bh = 0.125 * w, 0.25 * h, 0.5 * w, 0.5 * h
        kp = []
        for j in range(17):
            kp += [float(x + (j % 4) * bw / 4), float(y + (j // 4) * bh / 5), 2 if (i != 5 or j < 4) else 0]
        annotations.append({'id': ann_id, 'image_id': 100 + i, 'category_id': 1, 'iscrowd': 0, 'area': bw * bh,
                            'bbox': [x, y, bw, bh] if i != 6 else [x, y, 0.5, bh],     # image 6: degenerate box
                            'segmentation': [[x, y, x + bw, y, x + bw, y + bh, x, y + bh]],
                            'keypoints': kp, 'num_keypoints': sum(1 for v in kp[2::3] if v > 0)})
        ann_id += 1
        if i == 0:                                      # a crowd region is ignored by the detector targets
            annotations.append({'id': ann_id, 'image_id': 100, 'category_id': 1, 'iscrowd': 1, 'area': 4.0,
                                'bbox': [1, 1, 2, 2], 'segmentation': {'counts': [0, 4], 'size': [h, w]},
                                'keypoints': [0] * 51, 'num_keypoints': 0})
            ann_id += 1
    ann_file = os.path.join(root, 'person_keypoints_tiny.json')
    with open(ann_file, 'w') as fp:
        json.dump({'images': images, 'annotations': annotations,
                   'categories': [{'id': 1, 'name': 'person', 'keypoints': ['k%d' % j for j in range(17)]}]}, fp)
    return os.path.join(root, 'images'), ann_file
