"""Restatement of the pycocotools (2.0.x) pieces the reference's evaluator drives: ``COCO`` (index + lookups,
``annToRLE``), ``COCOeval`` for ``iouType`` 'bbox' / 'segm' / 'keypoints', and the ``mask`` module (RLE encode / decode /
area / toBbox / iou / merge / frPyObjects incl. the polygon rasteriser and the compressed-string codec of maskApi.c).

TEST INFRASTRUCTURE (see oracle/__init__.py).  pycocotools is a third-party dependency of the reference
(``Pipfile`` / ``src/utils/coco_eval_util.py:5-10``), absent from this image: PARITY UNPINNED for its arithmetic,
restated here from the published ``pycocotools/coco.py`` / ``cocoeval.py`` / ``_mask.pyx`` (bbIou) so that the
reference's OWN evaluator code -- ``CocoEvaluator``, its copies of ``loadRes`` / ``evaluate`` / ``createIndex``,
``convert_to_coco_api`` (``src/utils/coco_eval_util.py``) -- runs unmodified when tests/golden/make_golden.py writes
``tiny_coco_eval.npz``; the product evaluator (hnd_ghnd_object_detectors_amd/utils/coco_eval_util.py) is then checked
against those statistics.
"""
import copy
import datetime  # noqa: F401  (pycocotools imports it; kept for fidelity of the module surface)
from collections import defaultdict

import numpy as np


# ------------------------------------------------------------------------------------------------- _mask.iou on boxes
def bb_iou(dt, gt, iscrowd):
    """maskApi.c bbIou: dt [m][4], gt [n][4] in xywh; returns m x n; crowd ground truth: union = area(dt)"""
    dt = np.asarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    m, n = len(dt), len(gt)
    o = np.zeros((m, n), dtype=np.float64)
    for g in range(n):
        G = gt[g]
        ga = G[2] * G[3]
        crowd = iscrowd is not None and len(iscrowd) and iscrowd[g]
        for d in range(m):
            D = dt[d]
            da = D[2] * D[3]
            w = min(D[2] + D[0], G[2] + G[0]) - max(D[0], G[0])
            if w <= 0:
                continue
            h = min(D[3] + D[1], G[3] + G[1]) - max(D[1], G[1])
            if h <= 0:
                continue
            i = w * h
            u = da if crowd else da + ga - i
            o[d, g] = i / u
    return o


# ------------------------------------------------------------------------------------------------- maskApi.c (RLE)
# An RLE is {'size': [h, w], 'counts': bytes | str}: run lengths of the COLUMN-MAJOR mask, starting with a run of
# zeros, compressed by rleToString.  Internally: the list of run lengths.
def _rle_counts_of_mask(m):
    """rleEncode: m [h, w] (any memory order) -> run lengths over the column-major flattening"""
    flat = np.asarray(m, dtype=np.uint8).reshape(m.shape[0], m.shape[1]).flatten(order='F')
    if flat.size == 0:
        return [0]
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(bounds).tolist()
    if flat[0] != 0:
        runs = [0] + runs
    return [int(r) for r in runs]


def _rle_to_string(cnts):
    """rleToString: LEB128-like, 5 data bits + continuation bit per char, chars offset by 48; counts from the third
    on are stored as the difference to the count two places earlier"""
    out = bytearray()
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def _rle_from_string(s):
    if isinstance(s, str):
        s = s.encode('ascii')
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def _counts(rle):
    c = rle['counts']
    return [int(v) for v in c] if isinstance(c, (list, tuple, np.ndarray)) else _rle_from_string(c)


def _rle_of_counts(cnts, h, w):
    return {'size': [int(h), int(w)], 'counts': _rle_to_string(cnts)}


def _decode_one(rle):
    h, w = rle['size']
    cnts = _counts(rle)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for c in cnts:
        if v:
            flat[pos:pos + c] = 1
        pos += c
        v ^= 1
    return flat.reshape((h, w), order='F')


def _fr_poly(xy, h, w):
    """rleFrPoly: polygon [x0, y0, x1, y1, ...] -> run lengths.  Upsample x5, walk every edge densely, keep the
    points where x changes, downsample to pixel-column crossings, difference-encode."""
    k = len(xy) // 2
    scale = 5.0
    x = [int(scale * xy[2 * j] + .5) for j in range(k)]
    y = [int(scale * xy[2 * j + 1] + .5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            sl = (ye - ys) / dx if dx else 0.0       # dx == dy == 0: C divides 0/0 -> NaN cast; a single point
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + sl * t + .5))
        else:
            sl = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + sl * t + .5))
    xs_, ys_ = [], []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + .5) / scale - .5
            if np.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + .5) / scale - .5
            yd = 0.0 if yd < 0 else (float(h) if yd > h else yd)
            yd = np.ceil(yd)
            xs_.append(int(xd))
            ys_.append(int(yd))
    a = sorted([xx * int(h) + yy for xx, yy in zip(xs_, ys_)] + [int(h * w)])
    p = 0
    for j in range(len(a)):
        t = a[j]
        a[j] -= p
        p = t
    b, j = [a[0]], 1
    while j < len(a):
        if a[j] > 0:
            b.append(a[j])
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += a[j]
                j += 1
    return b


class _MaskModule(object):
    """pycocotools/mask.py over _mask.pyx"""

    @staticmethod
    def encode(bimask):
        bimask = np.asarray(bimask)
        if bimask.ndim == 3:
            h, w, n = bimask.shape
            return [_rle_of_counts(_rle_counts_of_mask(bimask[:, :, i]), h, w) for i in range(n)]
        h, w = bimask.shape
        return _rle_of_counts(_rle_counts_of_mask(bimask), h, w)

    @staticmethod
    def decode(rleObjs):
        if isinstance(rleObjs, list):
            return np.stack([_decode_one(r) for r in rleObjs], axis=2) if len(rleObjs) else np.zeros((0, 0, 0), np.uint8)
        return _decode_one(rleObjs)

    @staticmethod
    def area(rleObjs):
        def one(r):
            return int(sum(_counts(r)[1::2]))
        if isinstance(rleObjs, list):
            return np.array([one(r) for r in rleObjs], dtype=np.uint32)
        return np.uint32(one(rleObjs))

    @staticmethod
    def toBbox(rleObjs):
        def one(r):
            m = _decode_one(r)
            ys, xs = np.nonzero(m)
            if len(xs) == 0:
                return [0.0, 0.0, 0.0, 0.0]
            return [float(xs.min()), float(ys.min()), float(xs.max() - xs.min() + 1), float(ys.max() - ys.min() + 1)]
        if isinstance(rleObjs, list):
            return np.array([one(r) for r in rleObjs], dtype=np.double).reshape(-1, 4)
        return np.array(one(rleObjs), dtype=np.double)

    @staticmethod
    def merge(rleObjs, intersect=0):
        masks = [_decode_one(r) for r in rleObjs]
        h, w = rleObjs[0]['size']
        m = masks[0].astype(bool)
        for o in masks[1:]:
            m = (m & o.astype(bool)) if intersect else (m | o.astype(bool))
        return _rle_of_counts(_rle_counts_of_mask(m.astype(np.uint8)), h, w)

    @staticmethod
    def frPyObjects(pyobj, h, w):
        def fr_bbox(bb):
            xs, ys, xe, ye = bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]
            return _rle_of_counts(_fr_poly([xs, ys, xs, ye, xe, ye, xe, ys], h, w), h, w)
        if isinstance(pyobj, np.ndarray):
            return [fr_bbox(bb) for bb in pyobj.reshape(-1, 4)]
        if isinstance(pyobj, list) and len(pyobj) and isinstance(pyobj[0], dict):
            return [_rle_of_counts([int(c) for c in o['counts']], o['size'][0], o['size'][1]) for o in pyobj]
        if isinstance(pyobj, list) and len(pyobj) and len(pyobj[0]) == 4:
            return [fr_bbox(bb) for bb in pyobj]
        if isinstance(pyobj, list) and len(pyobj) and len(pyobj[0]) > 4:
            return [_rle_of_counts(_fr_poly([float(v) for v in poly], h, w), h, w) for poly in pyobj]
        if isinstance(pyobj, dict) and 'counts' in pyobj and 'size' in pyobj:
            return _rle_of_counts([int(c) for c in pyobj['counts']], pyobj['size'][0], pyobj['size'][1])
        raise Exception('input type is not supported.')

    @staticmethod
    def iou(dt, gt, iscrowd):
        if len(dt) == 0 or len(gt) == 0:
            return []
        if isinstance(dt[0], dict):                          # rleIou: crowd ground truth -> union = area(dt)
            D = [_decode_one(r).astype(bool) for r in dt]
            Gm = [_decode_one(r).astype(bool) for r in gt]
            o = np.zeros((len(D), len(Gm)), dtype=np.float64)
            for g, gm in enumerate(Gm):
                crowd = iscrowd is not None and len(iscrowd) and iscrowd[g]
                ga = int(gm.sum())
                for d, dm in enumerate(D):
                    da = int(dm.sum())
                    i = int((dm & gm).sum())
                    u = da if crowd else da + ga - i
                    o[d, g] = (i / u) if u > 0 else 0.0         # rleIou: u==0 cannot occur for i>0; 0/0 guarded as 0
            return o
        return bb_iou(dt, gt, iscrowd)


mask = _MaskModule()


# ------------------------------------------------------------------------------------------------- coco.py
class COCO(object):
    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = dict(), dict(), dict(), dict()
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            import json
            self.dataset = json.load(open(annotation_file, 'r'))
            assert type(self.dataset) == dict, 'annotation file format {} not supported'.format(type(self.dataset))
            self.createIndex()

    def createIndex(self):
        anns, cats, imgs = {}, {}, {}
        imgToAnns, catToImgs = defaultdict(list), defaultdict(list)
        if 'annotations' in self.dataset:
            for ann in self.dataset['annotations']:
                imgToAnns[ann['image_id']].append(ann)
                anns[ann['id']] = ann
        if 'images' in self.dataset:
            for img in self.dataset['images']:
                imgs[img['id']] = img
        if 'categories' in self.dataset:
            for cat in self.dataset['categories']:
                cats[cat['id']] = cat
        if 'annotations' in self.dataset and 'categories' in self.dataset:
            for ann in self.dataset['annotations']:
                catToImgs[ann['category_id']].append(ann['image_id'])
        self.anns, self.imgToAnns, self.catToImgs, self.imgs, self.cats = anns, imgToAnns, catToImgs, imgs, cats

    @staticmethod
    def _is_array_like(obj):
        return hasattr(obj, '__iter__') and hasattr(obj, '__len__')

    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
        imgIds = imgIds if self._is_array_like(imgIds) else [imgIds]
        catIds = catIds if self._is_array_like(catIds) else [catIds]
        if len(imgIds) == len(catIds) == len(areaRng) == 0:
            anns = self.dataset['annotations']
        else:
            if not len(imgIds) == 0:
                lists = [self.imgToAnns[imgId] for imgId in imgIds if imgId in self.imgToAnns]
                anns = [a for lst in lists for a in lst]
            else:
                anns = self.dataset['annotations']
            anns = anns if len(catIds) == 0 else [ann for ann in anns if ann['category_id'] in catIds]
            anns = anns if len(areaRng) == 0 else [ann for ann in anns
                                                   if ann['area'] > areaRng[0] and ann['area'] < areaRng[1]]
        if iscrowd is not None:
            return [ann['id'] for ann in anns if ann['iscrowd'] == iscrowd]
        return [ann['id'] for ann in anns]

    def getCatIds(self, catNms=[], supNms=[], catIds=[]):
        cats = self.dataset['categories']
        return [cat['id'] for cat in cats]

    def getImgIds(self, imgIds=[], catIds=[]):
        imgIds = imgIds if self._is_array_like(imgIds) else [imgIds]
        catIds = catIds if self._is_array_like(catIds) else [catIds]
        if len(imgIds) == len(catIds) == 0:
            ids = self.imgs.keys()
        else:
            ids = set(imgIds)
            for i, catId in enumerate(catIds):
                if i == 0 and len(ids) == 0:
                    ids = set(self.catToImgs[catId])
                else:
                    ids &= set(self.catToImgs[catId])
        return list(ids)

    def loadAnns(self, ids=[]):
        if self._is_array_like(ids):
            return [self.anns[i] for i in ids]
        if type(ids) == int:
            return [self.anns[ids]]

    def annToRLE(self, ann):
        t = self.imgs[ann['image_id']]
        h, w = t['height'], t['width']
        segm = ann['segmentation']
        if type(segm) == list:                              # polygons: one object may have several parts
            return mask.merge(mask.frPyObjects(segm, h, w))
        if type(segm['counts']) == list:                    # uncompressed RLE
            return mask.frPyObjects(segm, h, w)
        return ann['segmentation']

    def annToMask(self, ann):
        return mask.decode(self.annToRLE(ann))


# ------------------------------------------------------------------------------------------------- cocoeval.py
class Params(object):
    def setDetParams(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'small', 'medium', 'large']
        self.useCats = 1

    def setKpParams(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [20]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'medium', 'large']
        self.useCats = 1
        self.kpt_oks_sigmas = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87,
                                        .89, .89]) / 10.0

    def __init__(self, iouType='segm'):
        if iouType == 'segm' or iouType == 'bbox':
            self.setDetParams()
        elif iouType == 'keypoints':
            self.setKpParams()
        else:
            raise Exception('iouType not supported')
        self.iouType = iouType
        self.useSegm = None


class COCOeval(object):
    def __init__(self, cocoGt=None, cocoDt=None, iouType='segm'):
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.evalImgs = defaultdict(list)
        self.eval = {}
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        self.params = Params(iouType=iouType)
        self._paramsEval = {}
        self.stats = []
        self.ious = {}
        if cocoGt is not None:
            self.params.imgIds = sorted(cocoGt.getImgIds())
            self.params.catIds = sorted(cocoGt.getCatIds())

    def _prepare(self):
        p = self.params
        if p.useCats:
            gts = self.cocoGt.loadAnns(self.cocoGt.getAnnIds(imgIds=p.imgIds, catIds=p.catIds))
            dts = self.cocoDt.loadAnns(self.cocoDt.getAnnIds(imgIds=p.imgIds, catIds=p.catIds))
        else:
            gts = self.cocoGt.loadAnns(self.cocoGt.getAnnIds(imgIds=p.imgIds))
            dts = self.cocoDt.loadAnns(self.cocoDt.getAnnIds(imgIds=p.imgIds))
        if p.iouType == 'segm':                                 # _toMask: every annotation's segmentation -> RLE
            for anns, coco in ((gts, self.cocoGt), (dts, self.cocoDt)):
                for ann in anns:
                    ann['segmentation'] = coco.annToRLE(ann)
        for gt in gts:
            gt['ignore'] = gt['ignore'] if 'ignore' in gt else 0
            gt['ignore'] = 'iscrowd' in gt and gt['iscrowd']
            if p.iouType == 'keypoints':
                gt['ignore'] = (gt['num_keypoints'] == 0) or gt['ignore']
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for gt in gts:
            self._gts[gt['image_id'], gt['category_id']].append(gt)
        for dt in dts:
            self._dts[dt['image_id'], dt['category_id']].append(dt)
        self.evalImgs = defaultdict(list)
        self.eval = {}

    def computeIoU(self, imgId, catId):
        p = self.params
        if p.useCats:
            gt, dt = self._gts[imgId, catId], self._dts[imgId, catId]
        else:
            gt = [_ for cId in p.catIds for _ in self._gts[imgId, cId]]
            dt = [_ for cId in p.catIds for _ in self._dts[imgId, cId]]
        if len(gt) == 0 and len(dt) == 0:
            return []
        inds = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in inds]
        if len(dt) > p.maxDets[-1]:
            dt = dt[0:p.maxDets[-1]]
        if p.iouType == 'segm':
            g = [g['segmentation'] for g in gt]
            d = [d['segmentation'] for d in dt]
        elif p.iouType == 'bbox':
            g = [g['bbox'] for g in gt]
            d = [d['bbox'] for d in dt]
        else:
            raise Exception('unknown iouType for iou computation')
        iscrowd = [int(o['iscrowd']) for o in gt]
        return mask.iou(d, g, iscrowd)

    def computeOks(self, imgId, catId):
        p = self.params
        gts, dts = self._gts[imgId, catId], self._dts[imgId, catId]
        inds = np.argsort([-d['score'] for d in dts], kind='mergesort')
        dts = [dts[i] for i in inds]
        if len(dts) > p.maxDets[-1]:
            dts = dts[0:p.maxDets[-1]]
        if len(gts) == 0 or len(dts) == 0:
            return []
        ious = np.zeros((len(dts), len(gts)))
        sigmas = p.kpt_oks_sigmas
        vars = (sigmas * 2) ** 2
        k = len(sigmas)
        for j, gt in enumerate(gts):
            g = np.array(gt['keypoints'])
            xg, yg, vg = g[0::3], g[1::3], g[2::3]
            k1 = np.count_nonzero(vg > 0)
            bb = gt['bbox']
            x0, x1 = bb[0] - bb[2], bb[0] + bb[2] * 2
            y0, y1 = bb[1] - bb[3], bb[1] + bb[3] * 2
            for i, dt in enumerate(dts):
                d = np.array(dt['keypoints'])
                xd, yd = d[0::3], d[1::3]
                if k1 > 0:
                    dx, dy = xd - xg, yd - yg
                else:                                           # no labelled keypoint: distance to the doubled box
                    z = np.zeros((k))
                    dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
                    dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
                e = (dx ** 2 + dy ** 2) / vars / (gt['area'] + np.spacing(1)) / 2
                if k1 > 0:
                    e = e[vg > 0]
                ious[i, j] = np.sum(np.exp(-e)) / e.shape[0]
        return ious

    def evaluateImg(self, imgId, catId, aRng, maxDet):
        p = self.params
        if p.useCats:
            gt, dt = self._gts[imgId, catId], self._dts[imgId, catId]
        else:
            gt = [_ for cId in p.catIds for _ in self._gts[imgId, cId]]
            dt = [_ for cId in p.catIds for _ in self._dts[imgId, cId]]
        if len(gt) == 0 and len(dt) == 0:
            return None
        for g in gt:
            if g['ignore'] or (g['area'] < aRng[0] or g['area'] > aRng[1]):
                g['_ignore'] = 1
            else:
                g['_ignore'] = 0
        gtind = np.argsort([g['_ignore'] for g in gt], kind='mergesort')
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in dtind[0:maxDet]]
        iscrowd = [int(o['iscrowd']) for o in gt]
        ious = self.ious[imgId, catId][:, gtind] if len(self.ious[imgId, catId]) > 0 else self.ious[imgId, catId]
        T, G, D = len(p.iouThrs), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gtIg = np.array([g['_ignore'] for g in gt])
        dtIg = np.zeros((T, D))
        if not len(ious) == 0:
            for tind, t in enumerate(p.iouThrs):
                for dind, d in enumerate(dt):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gind, g in enumerate(gt):
                        if gtm[tind, gind] > 0 and not iscrowd[gind]:
                            continue
                        if m > -1 and gtIg[m] == 0 and gtIg[gind] == 1:
                            break
                        if ious[dind, gind] < iou:
                            continue
                        iou = ious[dind, gind]
                        m = gind
                    if m == -1:
                        continue
                    dtIg[tind, dind] = gtIg[m]
                    dtm[tind, dind] = gt[m]['id']
                    gtm[tind, m] = d['id']
        a = np.array([d['area'] < aRng[0] or d['area'] > aRng[1] for d in dt]).reshape((1, len(dt)))
        dtIg = np.logical_or(dtIg, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return {'image_id': imgId, 'category_id': catId, 'aRng': aRng, 'maxDet': maxDet,
                'dtIds': [d['id'] for d in dt], 'gtIds': [g['id'] for g in gt], 'dtMatches': dtm, 'gtMatches': gtm,
                'dtScores': [d['score'] for d in dt], 'gtIgnore': gtIg, 'dtIgnore': dtIg}

    def accumulate(self, p=None):
        if not self.evalImgs:
            print('Please run evaluate() first')
        if p is None:
            p = self.params
        p.catIds = p.catIds if p.useCats == 1 else [-1]
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds) if p.useCats else 1, len(p.areaRng), len(p.maxDets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        _pe = self._paramsEval
        catIds = _pe.catIds if _pe.useCats else [-1]
        setK, setA, setM, setI = set(catIds), set(map(tuple, _pe.areaRng)), set(_pe.maxDets), set(_pe.imgIds)
        k_list = [n for n, k in enumerate(p.catIds) if k in setK]
        m_list = [m for n, m in enumerate(p.maxDets) if m in setM]
        a_list = [n for n, a in enumerate(map(lambda x: tuple(x), p.areaRng)) if a in setA]
        i_list = [n for n, i in enumerate(p.imgIds) if i in setI]
        I0, A0 = len(_pe.imgIds), len(_pe.areaRng)
        for k, k0 in enumerate(k_list):
            Nk = k0 * A0 * I0
            for a, a0 in enumerate(a_list):
                Na = a0 * I0
                for m, maxDet in enumerate(m_list):
                    E = [self.evalImgs[Nk + Na + i] for i in i_list]
                    E = [e for e in E if e is not None]
                    if len(E) == 0:
                        continue
                    dtScores = np.concatenate([e['dtScores'][0:maxDet] for e in E])
                    inds = np.argsort(-dtScores, kind='mergesort')
                    dtScoresSorted = dtScores[inds]
                    dtm = np.concatenate([e['dtMatches'][:, 0:maxDet] for e in E], axis=1)[:, inds]
                    dtIg = np.concatenate([e['dtIgnore'][:, 0:maxDet] for e in E], axis=1)[:, inds]
                    gtIg = np.concatenate([e['gtIgnore'] for e in E])
                    npig = np.count_nonzero(gtIg == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dtIg))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dtIg))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        tp, fp = np.array(tp), np.array(fp)
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q, ss = np.zeros((R,)), np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr, q = pr.tolist(), q.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds = np.searchsorted(rc, p.recThrs, side='left')
                        try:
                            for ri, pi in enumerate(inds):
                                q[ri] = pr[pi]
                                ss[ri] = dtScoresSorted[pi]
                        except Exception:      # noqa  (pycocotools: index past the end -> stop filling)
                            pass
                        precision[t, :, k, a, m] = np.array(q)
                        scores[t, :, k, a, m] = np.array(ss)
        self.eval = {'params': p, 'counts': [T, R, K, A, M], 'precision': precision, 'recall': recall,
                     'scores': scores}

    def summarize(self):
        def _summarize(ap=1, iouThr=None, areaRng='all', maxDets=100):
            p = self.params
            iStr = ' {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}'
            titleStr = 'Average Precision' if ap == 1 else 'Average Recall'
            typeStr = '(AP)' if ap == 1 else '(AR)'
            iouStr = '{:0.2f}:{:0.2f}'.format(p.iouThrs[0], p.iouThrs[-1]) if iouThr is None \
                else '{:0.2f}'.format(iouThr)
            aind = [i for i, aRng in enumerate(p.areaRngLbl) if aRng == areaRng]
            mind = [i for i, mDet in enumerate(p.maxDets) if mDet == maxDets]
            if ap == 1:
                s = self.eval['precision']
                if iouThr is not None:
                    t = np.where(iouThr == p.iouThrs)[0]
                    s = s[t]
                s = s[:, :, :, aind, mind]
            else:
                s = self.eval['recall']
                if iouThr is not None:
                    t = np.where(iouThr == p.iouThrs)[0]
                    s = s[t]
                s = s[:, :, aind, mind]
            mean_s = -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])
            print(iStr.format(titleStr, typeStr, iouStr, areaRng, maxDets, mean_s))
            return mean_s

        if not self.eval:
            raise Exception('Please run accumulate() first')
        if self.params.iouType == 'keypoints':
            self.stats = np.array([_summarize(1, maxDets=20), _summarize(1, maxDets=20, iouThr=.5),
                                   _summarize(1, maxDets=20, iouThr=.75), _summarize(1, maxDets=20, areaRng='medium'),
                                   _summarize(1, maxDets=20, areaRng='large'), _summarize(0, maxDets=20),
                                   _summarize(0, maxDets=20, iouThr=.5), _summarize(0, maxDets=20, iouThr=.75),
                                   _summarize(0, maxDets=20, areaRng='medium'), _summarize(0, maxDets=20, areaRng='large')])
            return
        stats = np.zeros((12,))
        stats[0] = _summarize(1)
        stats[1] = _summarize(1, iouThr=.5, maxDets=self.params.maxDets[2])
        stats[2] = _summarize(1, iouThr=.75, maxDets=self.params.maxDets[2])
        stats[3] = _summarize(1, areaRng='small', maxDets=self.params.maxDets[2])
        stats[4] = _summarize(1, areaRng='medium', maxDets=self.params.maxDets[2])
        stats[5] = _summarize(1, areaRng='large', maxDets=self.params.maxDets[2])
        stats[6] = _summarize(0, maxDets=self.params.maxDets[0])
        stats[7] = _summarize(0, maxDets=self.params.maxDets[1])
        stats[8] = _summarize(0, maxDets=self.params.maxDets[2])
        stats[9] = _summarize(0, areaRng='small', maxDets=self.params.maxDets[2])
        stats[10] = _summarize(0, areaRng='medium', maxDets=self.params.maxDets[2])
        stats[11] = _summarize(0, areaRng='large', maxDets=self.params.maxDets[2])
        self.stats = stats

    def __str__(self):
        self.summarize()
