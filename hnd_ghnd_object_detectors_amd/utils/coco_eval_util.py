"""COCO evaluation for the validation path (role of the reference's src/utils/coco_eval_util.py:15-150,225-233).

The reference drives pycocotools' ``COCOeval`` (third-party, pinned 2.0.1 in Pipfile.lock, absent from this image:
PARITY UNPINNED for its arithmetic), with ``iou_types`` = bbox (+ segm for Mask R-CNN, + keypoints for Keypoint
R-CNN), and selects the checkpoint on ``coco_eval['bbox'].stats[0]`` (src/mimic_runner.py:94-100).  ``CocoEval``
restates COCOeval's published algorithm in numpy (host-side bookkeeping over at most 100 detections per image; not a
device workload):

  * similarity of detection vs ground truth: bbox -- IoU in xywh; segm -- IoU of the binary masks (predictions
    thresholded at 0.5 and run-length encoded on arrival, utils/mask_util.py); against a crowd region the union is the
    detection's own area; keypoints -- object keypoint similarity: mean over the labelled keypoints of
    exp(-d^2 / (2 s^2 k_i^2)) with s^2 = ground-truth area, k_i = 2 sigma_i (the 17 COCO person sigmas); for a ground
    truth without labelled keypoints the distance to its box doubled in size;
  * per image / category / area range: ground truths sorted non-ignored first (crowd, out-of-range area, and for
    keypoints ``num_keypoints == 0`` are ignored), detections by descending score (stable), greedy matching per IoU
    threshold 0.50:0.05:0.95 with the "prefer a non-ignored match" rule, unmatched detections outside the area range
    ignored;
  * accumulate: per category / area / maxDets, detections of all images merged by score (stable), cumulative
    TP / FP -> precision made monotone from the right, sampled at the 101 recall thresholds;
  * summarize: the twelve standard statistics (ten for keypoints: maxDets 20, no 'small' range);
    ``stats[0]`` = AP @[.50:.95 | all].
"""
import copy
from collections import defaultdict

import numpy as np
import torch

from . import mask_util, misc_util


class CocoGT(object):
    """ground truth in COCO terms: images, categories, annotations (bbox xywh, area, iscrowd, id)"""

    def __init__(self):
        self.images, self.categories, self.anns = {}, set(), defaultdict(list)
        self._next_id = 1

    def add(self, image_id, bbox_xywh, category_id, area=None, iscrowd=0, height=None, width=None, rle=None,
            keypoints=None):
        """rle: uint32 run lengths of the instance mask (mask_util.encode); keypoints: flat [x, y, v] * 17"""
        self.images[image_id] = {'id': image_id, 'height': height, 'width': width}
        self.categories.add(int(category_id))
        x, y, w, h = (float(v) for v in bbox_xywh)
        ann = {'id': self._next_id, 'image_id': image_id, 'category_id': int(category_id), 'bbox': [x, y, w, h],
               'area': float(w * h if area is None else area), 'iscrowd': int(iscrowd)}
        if rle is not None:
            ann['rle'] = rle
        if keypoints is not None:
            ann['keypoints'] = [float(v) for v in keypoints]
            ann['num_keypoints'] = sum(1 for v in ann['keypoints'][2::3] if v != 0)
        self.anns[image_id].append(ann)
        self._next_id += 1


def get_coco_api_from_dataset(dataset):
    """convert_to_coco_api of the reference (src/utils/coco_util.py:148-195): walk the dataset's targets (boxes xyxy,
    labels, area, iscrowd, image_id, optional masks / keypoints) into a ground-truth index; a dataset exposing
    ``.coco`` annotations directly is read as is (polygons / RLE rasterised like COCO.annToRLE)."""
    for _ in range(10):
        if isinstance(dataset, torch.utils.data.Subset):
            dataset = dataset.dataset
    gt = CocoGT()
    if hasattr(dataset, 'coco') and hasattr(dataset.coco, 'annotations'):
        for img_id in dataset.ids:
            info = dataset.coco.imgs[img_id]
            h, w = info.get('height'), info.get('width')
            gt.images[img_id] = {'id': img_id, 'height': h, 'width': w}
            for a in dataset.coco.annotations(img_id):
                rle = None
                if a.get('segmentation') and h and w:
                    rle = mask_util.encode(mask_util.segmentation_to_mask(a['segmentation'], h, w))
                kps = a.get('keypoints')
                gt.add(img_id, a['bbox'], a['category_id'], a.get('area'), a.get('iscrowd', 0), h, w, rle, kps)
                if kps is not None and 'num_keypoints' in a:
                    gt.anns[img_id][-1]['num_keypoints'] = int(a['num_keypoints'])
        return gt
    for idx in range(len(dataset)):
        img, targets = dataset[idx]
        image_id = int(targets['image_id'])
        boxes = targets['boxes'].clone().float().cpu()
        boxes[:, 2:] -= boxes[:, :2]
        n = len(boxes)
        areas = targets['area'].tolist() if 'area' in targets else [None] * n
        crowd = targets['iscrowd'].tolist() if 'iscrowd' in targets else [0] * n
        masks = targets['masks'].cpu().numpy() if 'masks' in targets else None
        kps = targets['keypoints'].reshape(n, -1).tolist() if 'keypoints' in targets and n else None
        gt.images[image_id] = {'id': image_id, 'height': int(img.shape[-2]), 'width': int(img.shape[-1])}
        for i, (b, l, a, c) in enumerate(zip(boxes.tolist(), targets['labels'].tolist(), areas, crowd)):
            gt.add(image_id, b, l, a, c, int(img.shape[-2]), int(img.shape[-1]),
                   mask_util.encode(masks[i]) if masks is not None else None, kps[i] if kps is not None else None)
    return gt


def get_iou_types(model):
    """models.get_iou_types (reference src/models/__init__.py:60-70): bbox, + segm for Mask R-CNN, + keypoints for
    Keypoint R-CNN"""
    from ..models import get_iou_types as by_class
    return by_class(model)


def bbox_iou(dt, gt, iscrowd):
    """pycocotools maskApi bbIou: dt [D, 4], gt [G, 4] in xywh -> [D, G]; crowd gt: union = area(dt)"""
    dt, gt = np.asarray(dt, dtype=np.float64).reshape(-1, 4), np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((len(dt), len(gt)))
    if len(dt) == 0 or len(gt) == 0:
        return out
    da, ga = dt[:, 2] * dt[:, 3], gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - \
        np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - \
        np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.where((w <= 0) | (h <= 0), 0.0, w * h)
    union = np.where(np.asarray(iscrowd, dtype=bool)[None, :], da[:, None], da[:, None] + ga[None, :] - inter)
    return inter / union


def mask_iou(dt, gt, iscrowd, h, w):
    """maskApi rleIou on run-length masks: dt / gt lists of uint32 run lengths of [h, w] masks -> [D, G]"""
    out = np.zeros((len(dt), len(gt)))
    if len(dt) == 0 or len(gt) == 0:
        return out
    D = [mask_util.decode(c, h, w) for c in dt]
    G = [mask_util.decode(c, h, w) for c in gt]
    da, ga = [int(m.sum()) for m in D], [int(m.sum()) for m in G]
    for j, g in enumerate(G):
        for i, d in enumerate(D):
            inter = int(np.count_nonzero(d & g))
            union = da[i] if iscrowd[j] else da[i] + ga[j] - inter
            out[i, j] = inter / union if union > 0 else 0.0
    return out


KPT_OKS_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def keypoint_oks(dt, gt):
    """COCOeval.computeOks: dt list of flat [x, y, v]*k, gt list of dicts (keypoints, bbox xywh, area) -> [D, G]"""
    out = np.zeros((len(dt), len(gt)))
    var = (KPT_OKS_SIGMAS * 2) ** 2
    k = len(KPT_OKS_SIGMAS)
    for j, g in enumerate(gt):
        kp = np.array(g['keypoints'])
        xg, yg, vg = kp[0::3], kp[1::3], kp[2::3]
        k1 = np.count_nonzero(vg > 0)
        bb = g['bbox']
        x0, x1, y0, y1 = bb[0] - bb[2], bb[0] + bb[2] * 2, bb[1] - bb[3], bb[1] + bb[3] * 2
        for i, d in enumerate(dt):
            d = np.array(d)
            xd, yd = d[0::3], d[1::3]
            if k1 > 0:
                dx, dy = xd - xg, yd - yg
            else:
                z = np.zeros(k)
                dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
                dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
            e = (dx ** 2 + dy ** 2) / var / (g['area'] + np.spacing(1)) / 2
            if k1 > 0:
                e = e[vg > 0]
            out[i, j] = np.sum(np.exp(-e)) / e.shape[0]
    return out


class CocoEval(object):
    """COCOeval(cocoGt, cocoDt, iou_type): evaluate() -> accumulate() -> summarize(); ``stats`` as pycocotools"""

    def __init__(self, gt, iou_type='bbox'):
        assert iou_type in ('bbox', 'segm', 'keypoints'), iou_type
        self.gt, self.iou_type = gt, iou_type
        self.iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        if iou_type == 'keypoints':                            # Params.setKpParams
            self.max_dets = [20]
            self.area_rng = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
            self.area_lbl = ['all', 'medium', 'large']
        else:                                                  # Params.setDetParams
            self.max_dets = [1, 10, 100]
            self.area_rng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
            self.area_lbl = ['all', 'small', 'medium', 'large']
        self.dts = defaultdict(list)          # (image_id, category_id) -> detections
        self.img_ids = []
        self.eval = None
        self.stats = np.zeros(10 if iou_type == 'keypoints' else 12)

    def add_detections(self, results):
        """results (loadRes): {'image_id', 'category_id', 'score'} + 'bbox' (xywh) | 'rle' (run lengths of the
        thresholded mask) | 'keypoints' (flat [x, y, v] * 17); the detection's area is the box area, the mask area,
        the extent of its keypoints"""
        for i, r in enumerate(results):
            d = dict(r)
            if self.iou_type == 'bbox':
                d['area'] = d['bbox'][2] * d['bbox'][3]
            elif self.iou_type == 'segm':
                d['area'] = mask_util.area(d['rle'])
            else:
                x, y = d['keypoints'][0::3], d['keypoints'][1::3]
                d['area'] = (np.max(x) - np.min(x)) * (np.max(y) - np.min(y))
            d['id'] = i + 1
            self.dts[(d['image_id'], d['category_id'])].append(d)

    def _similarity(self, img_id, dt, gt, iscrowd):
        if self.iou_type == 'bbox':
            return bbox_iou([d['bbox'] for d in dt], [g['bbox'] for g in gt], iscrowd)
        if self.iou_type == 'segm':
            info = self.gt.images[img_id]
            return mask_iou([d['rle'] for d in dt], [g['rle'] for g in gt], iscrowd, info['height'], info['width'])
        if len(dt) == 0 or len(gt) == 0:
            return np.zeros((len(dt), len(gt)))
        return keypoint_oks([d['keypoints'] for d in dt], gt)

    def _evaluate_img(self, img_id, cat_id, a_rng, max_det):
        gt = [dict(g) for g in self.gt.anns.get(img_id, []) if g['category_id'] == cat_id]
        dt = self.dts.get((img_id, cat_id), [])
        if len(gt) == 0 and len(dt) == 0:
            return None
        for g in gt:
            ignore = g['iscrowd'] or (self.iou_type == 'keypoints' and g.get('num_keypoints', 0) == 0)
            g['_ignore'] = 1 if (ignore or g['area'] < a_rng[0] or g['area'] > a_rng[1]) else 0
        gtind = np.argsort([g['_ignore'] for g in gt], kind='mergesort')
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in dtind[0:max_det]]
        iscrowd = [int(g['iscrowd']) for g in gt]
        # COCOeval.computeIoU: ONE similarity matrix per (image, category) over the detections sorted by score and
        # truncated to maxDets[-1] against the ground truth in annotation order; the four area ranges only permute its
        # columns (gtind), so it is cached instead of being recomputed (for segm: re-decoding every RLE) per range
        key = (img_id, cat_id)
        full = self._iou_cache.get(key)
        if full is None:
            gt0 = [g for g in self.gt.anns.get(img_id, []) if g['category_id'] == cat_id]
            dt_all = [self.dts.get(key, [])[i] for i in dtind[0:self.max_dets[-1]]]
            full = self._similarity(img_id, dt_all, gt0, [int(g['iscrowd']) for g in gt0])
            self._iou_cache[key] = full
        ious = full[:, gtind] if len(full) != 0 else full
        T, G, D = len(self.iou_thrs), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gt_ig = np.array([g['_ignore'] for g in gt])
        dt_ig = np.zeros((T, D))
        if len(ious) != 0:
            for tind, t in enumerate(self.iou_thrs):
                for dind in range(D):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gind in range(G):
                        if gtm[tind, gind] > 0 and not iscrowd[gind]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gind] == 1:
                            break
                        if ious[dind, gind] < iou:
                            continue
                        iou = ious[dind, gind]
                        m = gind
                    if m == -1:
                        continue
                    dt_ig[tind, dind] = gt_ig[m]
                    dtm[tind, dind] = gt[m]['id']
                    gtm[tind, m] = dt[dind]['id']
        a = np.array([d['area'] < a_rng[0] or d['area'] > a_rng[1] for d in dt]).reshape((1, len(dt)))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return {'dtMatches': dtm, 'dtScores': [d['score'] for d in dt], 'gtIgnore': gt_ig, 'dtIgnore': dt_ig}

    def evaluate(self, img_ids=None):
        self._iou_cache = {}
        self.img_ids = sorted(set(img_ids if img_ids is not None else self.gt.images.keys()))
        self.cat_ids = sorted(self.gt.categories)
        max_det = self.max_dets[-1]
        self.eval_imgs = [self._evaluate_img(i, c, a, max_det) for c in self.cat_ids for a in self.area_rng
                          for i in self.img_ids]

    def accumulate(self):
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(self.area_rng), len(self.max_dets)
        precision, recall = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
        I0 = len(self.img_ids)
        for k in range(K):
            for a in range(A):
                for m, max_det in enumerate(self.max_dets):
                    E = [self.eval_imgs[k * A * I0 + a * I0 + i] for i in range(I0)]
                    E = [e for e in E if e is not None]
                    if len(E) == 0:
                        continue
                    dt_scores = np.concatenate([e['dtScores'][0:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind='mergesort')
                    dtm = np.concatenate([e['dtMatches'][:, 0:max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e['dtIgnore'][:, 0:max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e['gtIgnore'] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q = np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        idx = np.searchsorted(rc, self.rec_thrs, side='left')
                        for ri, pi in enumerate(idx):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.eval = {'precision': precision, 'recall': recall}

    def _summarize(self, ap=1, iou_thr=None, area='all', max_dets=100):
        a, m = self.area_lbl.index(area), self.max_dets.index(max_dets)
        s = self.eval['precision'] if ap == 1 else self.eval['recall']
        if iou_thr is not None:
            s = s[np.where(np.isclose(iou_thr, self.iou_thrs))[0]]
        s = s[:, :, :, a, m] if ap == 1 else s[:, :, a, m]
        mean_s = -1 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))
        title, typ = ('Average Precision', '(AP)') if ap == 1 else ('Average Recall', '(AR)')
        iou_str = '{:0.2f}:{:0.2f}'.format(self.iou_thrs[0], self.iou_thrs[-1]) if iou_thr is None \
            else '{:0.2f}'.format(iou_thr)
        print(' {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}'.format(title, typ, iou_str, area,
                                                                                          max_dets, mean_s))
        return mean_s

    def summarize(self):
        S = self._summarize
        if self.iou_type == 'keypoints':
            self.stats = np.array([S(1, max_dets=20), S(1, .5, max_dets=20), S(1, .75, max_dets=20),
                                   S(1, area='medium', max_dets=20), S(1, area='large', max_dets=20),
                                   S(0, max_dets=20), S(0, .5, max_dets=20), S(0, .75, max_dets=20),
                                   S(0, area='medium', max_dets=20), S(0, area='large', max_dets=20)])
            return self.stats
        self.stats = np.array([S(1), S(1, .5), S(1, .75), S(1, area='small'), S(1, area='medium'), S(1, area='large'),
                               S(0, max_dets=1), S(0, max_dets=10), S(0), S(0, area='small'), S(0, area='medium'),
                               S(0, area='large')])
        return self.stats


def BBoxEval(gt):
    return CocoEval(gt, 'bbox')


class CocoEvaluator(object):
    """same surface as the reference's class: update(res) per batch, synchronize_between_processes(), accumulate(),
    summarize(); ``coco_eval['bbox'].stats[0]`` is the validation mAP mimic_runner compares."""

    def __init__(self, coco_gt, iou_types):
        assert isinstance(iou_types, (list, tuple))
        self.coco_gt = copy.deepcopy(coco_gt)
        self.iou_types = list(iou_types)
        self.coco_eval = {t: CocoEval(self.coco_gt, t) for t in self.iou_types}
        self.img_ids = []
        self.results = {t: [] for t in self.iou_types}

    def update(self, predictions):
        have = set(self.img_ids)          # a sampler-padded repeat of an image already scored here is dropped whole
        predictions = {k: v for k, v in predictions.items() if k not in have}
        self.img_ids.extend(sorted(predictions.keys()))
        for t in self.iou_types:
            self.results[t].extend(self.prepare(predictions, t))

    def prepare(self, predictions, iou_type):
        if iou_type == 'bbox':
            return self.prepare_for_coco_detection(predictions)
        if iou_type == 'segm':
            return self.prepare_for_coco_segmentation(predictions)
        if iou_type == 'keypoints':
            return self.prepare_for_coco_keypoint(predictions)
        raise ValueError('Unknown iou type {}'.format(iou_type))

    @staticmethod
    def prepare_for_coco_detection(predictions):
        out = []
        for image_id, pred in predictions.items():
            if len(pred) == 0 or len(pred['boxes']) == 0:
                continue
            boxes = pred['boxes'].detach().cpu().clone().float()
            boxes[:, 2:] -= boxes[:, :2]                                   # xyxy -> xywh (convert_to_xywh)
            for b, s, l in zip(boxes.tolist(), pred['scores'].tolist(), pred['labels'].tolist()):
                out.append({'image_id': image_id, 'category_id': l, 'bbox': b, 'score': s})
        return out

    @staticmethod
    def prepare_for_coco_segmentation(predictions):
        """masks [n, 1, H, W] probabilities -> thresholded at 0.5 (reference :101) -> run lengths"""
        out = []
        for image_id, pred in predictions.items():
            if len(pred) == 0 or len(pred['scores']) == 0:
                continue
            masks = pred['masks'][:, 0]
            if masks.is_cuda and masks.is_floating_point():     # threshold + run boundaries in one HIP launch
                rles = mask_util.encode_probs(masks, 0.5)
            else:
                rles = mask_util.encode_batch(masks > 0.5)
            for rle, s, l in zip(rles, pred['scores'].tolist(), pred['labels'].tolist()):
                out.append({'image_id': image_id, 'category_id': l, 'rle': rle, 'score': s})
        return out

    @staticmethod
    def prepare_for_coco_keypoint(predictions):
        out = []
        for image_id, pred in predictions.items():
            if len(pred) == 0 or len(pred['scores']) == 0:
                continue
            kps = pred['keypoints'].detach().cpu().flatten(start_dim=1).tolist()
            for k, s, l in zip(kps, pred['scores'].tolist(), pred['labels'].tolist()):
                out.append({'image_id': image_id, 'category_id': l, 'keypoints': k, 'score': s})
        return out

    def synchronize_between_processes(self):
        gathered = misc_util.all_gather((self.img_ids, self.results))
        # keep one copy per IMAGE (DistributedSampler pads the last shards with repeats): like the reference's merge
        # (np.unique(img_ids, return_index=True)) the first rank that reports an image id owns all its detections
        owner = {}
        for rank, (ids, _) in enumerate(gathered):
            for i in ids:
                owner.setdefault(i, rank)
        self.img_ids = sorted(owner)
        for t, ev in self.coco_eval.items():
            uniq = [r for rank, (_, res) in enumerate(gathered) for r in res[t] if owner.get(r['image_id']) == rank]
            self.results[t] = uniq
            ev.dts = defaultdict(list)
            ev.add_detections(uniq)
            ev.evaluate(self.img_ids)

    def accumulate(self):
        for ev in self.coco_eval.values():
            ev.accumulate()

    def summarize(self):
        for iou_type, ev in self.coco_eval.items():
            print('IoU metric: {}'.format(iou_type))
            ev.summarize()
