"""CPU: the bbox COCO evaluator of the validation path (utils/coco_eval_util.py) on hand-computable cases.
pycocotools (the reference's evaluator, src/utils/coco_eval_util.py) is not installable here, so these are the
known-answer cases of its published algorithm: 10 IoU thresholds, 101 recall thresholds, precision envelope."""
import numpy as np
import pytest
import torch

from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import BBoxEval, CocoEvaluator, CocoGT, bbox_iou


def _gt(boxes):
    """boxes: list of (image_id, [x, y, w, h], category, iscrowd)"""
    gt = CocoGT()
    for image_id, b, c, crowd in boxes:
        gt.add(image_id, b, c, iscrowd=crowd, height=200, width=200)
    return gt


def _run(gt, dets, img_ids=None):
    ev = BBoxEval(gt)
    ev.add_detections([{'image_id': i, 'category_id': c, 'bbox': b, 'score': s} for i, b, c, s in dets])
    ev.evaluate(img_ids)
    ev.accumulate()
    return ev.summarize()


def test_iou_and_crowd_union():
    iou = bbox_iou([[0, 0, 10, 10]], [[5, 0, 10, 10], [0, 0, 100, 100]], [0, 1])
    assert abs(iou[0, 0] - 50.0 / 150.0) < 1e-12
    assert abs(iou[0, 1] - 1.0) < 1e-12            # crowd: intersection / area(detection)


def test_perfect_detections_score_one(capsys):
    gt = _gt([(1, [10, 10, 50, 60], 1, 0), (1, [100, 20, 40, 40], 2, 0), (2, [5, 5, 120, 100], 1, 0)])
    stats = _run(gt, [(1, [10, 10, 50, 60], 1, 0.9), (1, [100, 20, 40, 40], 2, 0.8), (2, [5, 5, 120, 100], 1, 0.7)])
    assert all(abs(stats[i] - 1.0) < 1e-12 for i in (0, 1, 2, 8))
    out = capsys.readouterr().out
    assert 'Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=100 ] = 1.000' in out


def test_single_detection_at_iou_0_6():
    # IoU = 60/100: counts as a match at thresholds 0.50, 0.55, 0.60 only -> AP = 3/10, AP50 = 1, AP75 = 0
    gt = _gt([(1, [0, 0, 100, 100], 1, 0)])
    stats = _run(gt, [(1, [0, 0, 60, 100], 1, 0.5)])
    assert abs(stats[0] - 0.3) < 1e-12 and abs(stats[1] - 1.0) < 1e-12 and stats[2] == 0.0


def test_one_true_positive_one_false_positive_two_ground_truths():
    # recall reaches 0.5 at precision 1 -> 51 of the 101 recall thresholds (0.00 .. 0.50) score 1, the rest 0
    gt = _gt([(1, [0, 0, 50, 50], 1, 0), (1, [100, 100, 50, 50], 1, 0)])
    stats = _run(gt, [(1, [0, 0, 50, 50], 1, 0.9), (1, [60, 60, 20, 20], 1, 0.8)])
    assert abs(stats[0] - 51.0 / 101.0) < 1e-12
    # the false positive first: precision 0.5 at recall 0.5
    stats = _run(gt, [(1, [0, 0, 50, 50], 1, 0.7), (1, [60, 60, 20, 20], 1, 0.8)])
    assert abs(stats[0] - 0.5 * 51.0 / 101.0) < 1e-12


def test_crowd_matches_are_ignored_not_false_positives():
    gt = _gt([(1, [0, 0, 50, 50], 1, 0), (1, [100, 100, 80, 80], 1, 1)])
    dets = [(1, [0, 0, 50, 50], 1, 0.9), (1, [110, 110, 20, 20], 1, 0.95), (1, [120, 120, 30, 30], 1, 0.85)]
    stats = _run(gt, dets)                          # both detections inside the crowd region are ignored
    assert abs(stats[0] - 1.0) < 1e-12


def test_area_ranges_and_max_dets():
    gt = _gt([(1, [0, 0, 20, 20], 1, 0), (1, [50, 50, 120, 120], 1, 0)])       # one small (400), one large (14400)
    stats = _run(gt, [(1, [0, 0, 20, 20], 1, 0.6), (1, [50, 50, 120, 120], 1, 0.9)])
    assert abs(stats[3] - 1.0) < 1e-12 and abs(stats[5] - 1.0) < 1e-12 and stats[4] == -1     # no medium objects
    assert stats[6] == 0.5 and stats[7] == 1.0                                  # AR@1 sees only the top detection


def test_evaluator_surface_and_image_without_predictions():
    gt = _gt([(7, [10, 10, 40, 40], 3, 0), (8, [20, 20, 60, 30], 3, 0)])
    ev = CocoEvaluator(gt, ['bbox'])
    ev.update({7: {'boxes': torch.tensor([[10.0, 10.0, 50.0, 50.0]]), 'labels': torch.tensor([3]),
                   'scores': torch.tensor([0.8])}})
    ev.update({8: {'boxes': torch.zeros(0, 4), 'labels': torch.zeros(0, dtype=torch.int64), 'scores': torch.zeros(0)}})
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize()
    assert abs(ev.coco_eval['bbox'].stats[0] - 51.0 / 101.0) < 1e-12          # 1 of 2 ground truths found
    with pytest.raises(AssertionError):
        CocoEvaluator(gt, ['bbox', 'caption'])


def test_padded_sampler_repeats_are_dropped_by_image_id(monkeypatch):
    """DistributedSampler pads the last shards with repeats; the reference's merge keeps the FIRST rank's
    copy of each image id (np.unique(..., return_index=True)) -- by image, not by detection content: a repeat whose
    detections differ slightly (a different batch composition) must not count as extra false positives."""
    from hnd_ghnd_object_detectors_amd.utils import misc_util
    gt = _gt([(7, [10, 10, 40, 40], 3, 0), (8, [20, 20, 60, 30], 3, 0)])

    def pred(box, score):
        return {'boxes': torch.tensor([box]), 'labels': torch.tensor([3]), 'scores': torch.tensor([score])}
    rank0, rank1 = CocoEvaluator(gt, ['bbox']), CocoEvaluator(gt, ['bbox'])
    rank0.update({7: pred([10.0, 10.0, 50.0, 50.0], 0.8)})
    rank0.update({7: pred([11.0, 10.0, 50.0, 50.0], 0.7)})           # a repeat inside one shard: dropped at update()
    rank1.update({8: pred([20.0, 20.0, 80.0, 50.0], 0.9)})
    rank1.update({7: pred([10.5, 10.0, 50.0, 50.0], 0.79)})          # rank 1's padded repeat of image 7
    assert rank0.img_ids == [7] and len(rank0.results['bbox']) == 1
    monkeypatch.setattr(misc_util, 'all_gather', lambda _: [(rank0.img_ids, rank0.results),
                                                            (rank1.img_ids, rank1.results)])
    rank0.synchronize_between_processes()
    assert rank0.img_ids == [7, 8] and len(rank0.results['bbox']) == 2
    assert sorted((r['image_id'], r['score']) for r in rank0.results['bbox']) == [(7, 0.800000011920929), (8, 0.8999999761581421)]
    rank0.accumulate()
    rank0.summarize()
    assert abs(rank0.coco_eval['bbox'].stats[0] - 1.0) < 1e-12       # both ground truths found, no false positive


def test_ground_truth_from_a_dataset_of_targets():
    from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import get_coco_api_from_dataset

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 2

        def __getitem__(self, i):
            return torch.zeros(3, 50, 60), {'image_id': torch.tensor([i + 1]), 'labels': torch.tensor([1, 2]),
                                            'boxes': torch.tensor([[1.0, 2.0, 11.0, 22.0], [5.0, 5.0, 9.0, 9.0]]),
                                            'area': torch.tensor([200.0, 16.0]), 'iscrowd': torch.tensor([0, 1])}
    gt = get_coco_api_from_dataset(torch.utils.data.Subset(DS(), [0, 1]))
    assert sorted(gt.images) == [1, 2] and sorted(gt.categories) == [1, 2]
    a = gt.anns[1][0]
    assert a['bbox'] == [1.0, 2.0, 10.0, 20.0] and a['area'] == 200.0 and gt.anns[1][1]['iscrowd'] == 1


def test_evaluator_matches_the_reference_evaluator_fixture():
    """tiny_coco_eval.npz = the twelve statistics the REFERENCE's CocoEvaluator (its own loadRes / evaluate /
    createIndex copies, convert_to_coco_api) produced for seeded ground truth + predictions (crowd boxes, a wrong
    class, a duplicate, false positives; all three area ranges populated), over the restated pycocotools of
    oracle/pycoco_r.py.  The product evaluator must reproduce them."""
    from tests import golden_util as G
    from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import get_coco_api_from_dataset
    z = G.load_raw('tiny_coco_eval')
    dataset, preds = G.coco_eval_case_inputs(int(z['seed']))

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(dataset)

        def __getitem__(self, i):
            img, t = dataset[i]
            return img, {k: v.clone() for k, v in t.items()}
    gt = get_coco_api_from_dataset(DS())
    ev = CocoEvaluator(gt, ['bbox'])
    ids = sorted(preds)
    ev.update({i: preds[i] for i in ids[:4]})
    ev.update({i: preds[i] for i in ids[4:]})
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize()
    got, ref = ev.coco_eval['bbox'].stats, z['stats']
    assert np.abs(got - ref).max() < 1e-12, (got, ref)
    assert abs(float(np.clip(ev.coco_eval['bbox'].eval['precision'], 0, None).sum()) - float(z['precision_checksum'])) < 1e-9


def test_restated_pycocotools_agrees_on_the_hand_cases():
    """oracle/pycoco_r.COCOeval (what the reference evaluator ran over) and the product BBoxEval are two independent
    restatements of pycocotools: they must agree on the known-answer cases above as well"""
    from oracle.pycoco_r import COCO, COCOeval
    gt_boxes = [(1, [0, 0, 50, 50], 1, 0), (1, [100, 100, 50, 50], 1, 0), (1, [20, 20, 90, 90], 2, 1)]
    dets = [(1, [0, 0, 50, 50], 1, 0.9), (1, [60, 60, 20, 20], 1, 0.8), (1, [30, 30, 40, 40], 2, 0.7)]
    mine = _run(_gt(gt_boxes), dets)
    c = COCO()
    c.dataset = {'images': [{'id': 1}], 'categories': [{'id': 1}, {'id': 2}],
                 'annotations': [{'id': k + 1, 'image_id': i, 'bbox': b, 'category_id': cat, 'iscrowd': crowd,
                                  'area': float(b[2] * b[3])} for k, (i, b, cat, crowd) in enumerate(gt_boxes)]}
    c.createIndex()
    d = COCO()
    d.dataset = {'images': [{'id': 1}], 'categories': [{'id': 1}, {'id': 2}],
                 'annotations': [{'id': k + 1, 'image_id': i, 'bbox': b, 'category_id': cat, 'score': s, 'iscrowd': 0,
                                  'area': float(b[2] * b[3])} for k, (i, b, cat, s) in enumerate(dets)]}
    d.createIndex()
    e = COCOeval(c, d, 'bbox')
    p = e.params
    e._prepare()
    e.ious = {(i, k): e.computeIoU(i, k) for i in p.imgIds for k in p.catIds}
    e.evalImgs = [e.evaluateImg(i, k, a, p.maxDets[-1]) for k in p.catIds for a in p.areaRng for i in p.imgIds]
    import copy
    e._paramsEval = copy.deepcopy(p)
    e.accumulate()
    e.summarize()
    assert np.abs(np.asarray(e.stats) - mine).max() < 1e-12


# ------------------------------------------------------------------------------------------ segm / keypoints
def _evaluate_case(kind):
    from tests import golden_util as G
    from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import get_coco_api_from_dataset
    z = G.load_raw('tiny_coco_eval_' + kind)
    dataset, preds = G.coco_eval_case_inputs(int(z['seed']))
    dataset, preds = G.coco_eval_case_extras(dataset, preds, kind)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(dataset)

        def __getitem__(self, i):
            img, t = dataset[i]
            return img, {k: v.clone() for k, v in t.items()}
    ev = CocoEvaluator(get_coco_api_from_dataset(DS()), ['bbox', kind])
    ids = sorted(preds)
    ev.update({i: preds[i] for i in ids[:3]})
    ev.update({i: preds[i] for i in ids[3:]})
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize()
    return z, ev


@pytest.mark.parametrize('kind', ['segm', 'keypoints'])
def test_segm_and_keypoint_evaluation_match_the_reference_evaluator_fixtures(kind):
    """tiny_coco_eval_{segm,keypoints}.npz: the REFERENCE's CocoEvaluator with iou_types bbox + segm / bbox + keypoints
    (what src/models/__init__.py:60-70 selects for Mask / Keypoint R-CNN) -- prepare_for_coco_segmentation's 0.5
    threshold and RLE encoding, loadRes's mask areas / keypoint extents, COCOeval's mask IoU / OKS -- over the
    restated pycocotools.  The product evaluator (binary-mask IoU on its own run lengths, OKS) must reproduce the
    statistics of both metrics."""
    z, ev = _evaluate_case(kind)
    assert np.abs(ev.coco_eval['bbox'].stats - z['stats']).max() < 1e-12
    got, ref = ev.coco_eval[kind].stats, z['stats_' + kind]
    assert got.shape == ref.shape == ((12,) if kind == 'segm' else (10,))
    assert np.abs(got - ref).max() < 1e-12, (got, ref)
    chk = float(np.clip(ev.coco_eval[kind].eval['precision'], 0, None).sum())
    assert abs(chk - float(z['precision_checksum_' + kind])) < 1e-9


def test_mask_util_matches_the_restated_mask_api():
    """utils/mask_util.py (vectorised) vs oracle/pycoco_r.py (maskApi.c line by line): polygon rasteriser, run-length
    codec, compressed strings -- on random polygons incl. integer vertices, vertices outside the image, repeated
    vertices -- and the two documented conventions: a w x h box covers exactly w*h pixels, runs are column-major"""
    from oracle import pycoco_r as P
    from hnd_ghnd_object_detectors_amd.utils import mask_util as MU
    rng = np.random.RandomState(5)
    for t in range(120):
        h, w = rng.randint(5, 70), rng.randint(5, 70)
        pts = rng.rand(rng.randint(3, 9), 2) * [w * 1.3, h * 1.3] - [w * 0.15, h * 0.15]
        if t % 3 == 0:
            pts = np.round(pts)
        if t % 7 == 0:
            pts[1] = pts[0]
        poly = pts.flatten().tolist()
        ref = P.mask.decode(P.mask.frPyObjects([poly], h, w)[0]).astype(bool)
        got = MU.polygons_to_mask([poly], h, w)
        assert (ref == got).all(), t
        rle = P.mask.encode(np.asfortranarray(ref.astype(np.uint8)))
        counts = MU.encode(got)
        assert MU.counts_to_string(counts) == rle['counts']
        assert (MU.string_to_counts(rle['counts'].decode('ascii')) == counts).all()
        assert (MU.decode(counts, h, w) == got).all() and MU.area(counts) == int(got.sum()) == int(P.mask.area(rle))
    assert int(MU.bbox_to_mask([2, 1, 5, 4], 8, 10).sum()) == 20
    stack = torch.from_numpy(rng.rand(7, 23, 31) > 0.6)
    stack[2] = False
    stack[3] = True
    for got, m in zip(MU.encode_batch(stack), stack.numpy()):
        assert got.dtype == np.uint32 and got.tolist() == MU.encode(m).tolist()
    assert MU.encode_batch(torch.zeros(0, 4, 4, dtype=torch.bool)) == []
    m = np.zeros((3, 4), dtype=bool)
    m[1:, 0] = True                                          # column-major: 1 zero, 2 ones, 9 zeros
    assert MU.encode(m).tolist() == [1, 2, 9]
    two_parts = MU.polygons_to_mask([[0, 0, 0, 2, 2, 2, 2, 0], [3, 3, 3, 5, 5, 5, 5, 3]], 6, 6)
    assert int(two_parts.sum()) == 8                          # parts of one object are OR-ed (annToRLE merge)
    assert (MU.segmentation_to_mask({'counts': [1, 2, 9], 'size': [3, 4]}, 3, 4) == m).all()
    assert (MU.segmentation_to_mask({'counts': MU.counts_to_string([1, 2, 9]), 'size': [3, 4]}, 3, 4) == m).all()


def test_keypoint_similarity_known_answers():
    """OKS by hand: identical keypoints -> 1; every labelled keypoint displaced by d -> mean exp(-d^2 / (2 area
    (2 sigma_i)^2)); a ground truth without labelled keypoints scores 1 while the detection stays inside its box
    doubled in size; such a ground truth is ignored by the evaluation (num_keypoints == 0)."""
    from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import keypoint_oks, KPT_OKS_SIGMAS, CocoEval
    kp = np.zeros(51)
    kp[0::3], kp[1::3], kp[2::3] = np.arange(17) * 3.0 + 20, np.arange(17) * 2.0 + 30, 2
    g = {'keypoints': kp.tolist(), 'bbox': [20.0, 30.0, 50.0, 34.0], 'area': 1700.0}
    assert abs(keypoint_oks([kp.tolist()], [g])[0, 0] - 1.0) < 1e-15
    moved = kp.copy()
    moved[0::3] += 4.0
    want = np.mean(np.exp(-16.0 / (2 * (1700.0 + np.spacing(1)) * (2 * KPT_OKS_SIGMAS) ** 2)))
    assert abs(keypoint_oks([moved.tolist()], [g])[0, 0] - want) < 1e-15
    unl = dict(g, keypoints=(kp * np.tile([1, 1, 0], 17)).tolist())
    assert abs(keypoint_oks([moved.tolist()], [unl])[0, 0] - 1.0) < 1e-15
    gt = CocoGT()
    gt.add(1, g['bbox'], 1, g['area'], 0, 100, 100, keypoints=g['keypoints'])
    gt.add(1, [60.0, 10.0, 20.0, 20.0], 1, 400.0, 0, 100, 100, keypoints=unl['keypoints'])
    assert [a['num_keypoints'] for a in gt.anns[1]] == [17, 0]
    ev = CocoEval(gt, 'keypoints')
    ev.add_detections([{'image_id': 1, 'category_id': 1, 'keypoints': kp.tolist(), 'score': 0.9}])
    ev.evaluate()
    ev.accumulate()
    stats = ev.summarize()
    assert stats.shape == (10,) and abs(stats[0] - 1.0) < 1e-12 and abs(stats[5] - 1.0) < 1e-12


def test_mask_iou_and_crowd_known_answers():
    from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import mask_iou
    from hnd_ghnd_object_detectors_amd.utils import mask_util as MU
    a = np.zeros((10, 10), dtype=bool)
    b = np.zeros((10, 10), dtype=bool)
    a[:4, :5] = True                                          # 20 px
    b[2:6, :5] = True                                         # 20 px, 10 shared
    iou = mask_iou([MU.encode(a)], [MU.encode(b), MU.encode(b)], [0, 1], 10, 10)
    assert abs(iou[0, 0] - 10.0 / 30.0) < 1e-15 and abs(iou[0, 1] - 10.0 / 20.0) < 1e-15
    assert mask_iou([MU.encode(np.zeros((10, 10), bool))], [MU.encode(b)], [0], 10, 10)[0, 0] == 0.0


def test_ground_truth_from_a_coco_annotation_file(tmp_path):
    """a dataset exposing ``.coco`` annotations (reference: get_coco_api_from_dataset returns dataset.coco for
    CocoDetection datasets): polygons and the crowd region's uncompressed RLE become masks like COCO.annToRLE, keypoints
    and num_keypoints are carried; perfect predictions then score AP 1 on all three metrics"""
    from tests.coco_fixture import write_tiny_coco
    from hnd_ghnd_object_detectors_amd.utils import coco_util, data_util, mask_util
    from hnd_ghnd_object_detectors_amd.utils.coco_eval_util import get_coco_api_from_dataset
    img_dir, ann_file = write_tiny_coco(str(tmp_path))
    ds = coco_util.get_coco(img_dir, ann_file, data_util.get_transform(False), remove_non_annotated_imgs=False)
    gt = get_coco_api_from_dataset(ds)
    person, crowd = gt.anns[100]
    assert crowd['iscrowd'] == 1 and mask_util.area(crowd['rle']) == 4 and crowd['num_keypoints'] == 0
    m = mask_util.decode(person['rle'], 48, 64)
    assert int(m.sum()) == 32 * 24 and m[12:36, 8:40].all() and person['num_keypoints'] == 17
    assert gt.anns[105][0]['num_keypoints'] == 4
    preds = {}
    for img_id, anns in gt.anns.items():
        anns = [a for a in anns if not a['iscrowd']]
        info = gt.images[img_id]
        boxes = torch.tensor([[a['bbox'][0], a['bbox'][1], a['bbox'][0] + a['bbox'][2], a['bbox'][1] + a['bbox'][3]]
                              for a in anns])
        masks = torch.stack([torch.from_numpy(mask_util.decode(a['rle'], info['height'], info['width'])).float()
                             for a in anns])[:, None]
        kps = torch.tensor([a['keypoints'] for a in anns]).view(len(anns), 17, 3)
        preds[img_id] = {'boxes': boxes, 'labels': torch.ones(len(anns), dtype=torch.int64),
                         'scores': torch.full((len(anns),), 0.9), 'masks': masks, 'keypoints': kps}
    ev = CocoEvaluator(gt, ['bbox', 'segm', 'keypoints'])
    ev.update(preds)
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize()
    for kind in ('bbox', 'segm', 'keypoints'):
        assert abs(ev.coco_eval[kind].stats[0] - 1.0) < 1e-12, kind
