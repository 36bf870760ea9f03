"""Helpers to read tests/golden/*.npz (written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ZERO_GRAD_SUFFIXES = ('layer1.decoder.3.bias', 'layer1.decoder.8.bias')


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    return z, meta


def load_raw(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)


def pipeline_case(z):
    """(uint8 HWC image, flip flag, targets before, targets after) per image of tiny_input_pipeline.npz"""
    out = []
    n = len([k for k in z.files if k.startswith('u8/')])
    for i in range(n):
        tin = {k: torch.from_numpy(z['target_in/%d/%s' % (i, k)]) for k in ('boxes', 'masks', 'keypoints')}
        tout = {k: torch.from_numpy(z['target_out/%d/%s' % (i, k)]) for k in ('boxes', 'masks', 'keypoints')}
        out.append((torch.from_numpy(z['u8/%d' % i]), bool(z['flip/%d' % i]), tin, tout))
    return out


def ext_case_inputs(meta):
    """Re-create the seeded inputs of make_golden.run_ext_filter_case (kept in sync by test_oracle_golden)."""
    g = torch.Generator().manual_seed(1234 + meta['seed'])
    images, targets = [], []
    for i, (h, w) in enumerate(meta['sizes']):
        images.append(torch.rand(3, h, w, generator=g))
        kp = torch.rand(1, 17, 3, generator=g) * torch.tensor([w, h, 1.0])
        kp[..., 2] = 1.0 if i != 2 else 0.0
        if i == 2:
            kp[0, :5, 2] = 1.0
        box = [[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]] if i != 1 else [[3.0, 4.0, 0.5, 20.0]]
        targets.append({'boxes': torch.tensor(box), 'labels': torch.tensor([1]), 'keypoints': kp})
    return images, targets


def case_inputs(meta):
    """Re-create the seeded inputs of make_golden.make_inputs (kept in sync by test_oracle_golden)."""
    g = torch.Generator().manual_seed(1234 + meta['seed'])
    images, targets = [], []
    for h, w in meta['sizes']:
        images.append(torch.rand(3, h, w, generator=g))
        t = {'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]]), 'labels': torch.tensor([1])}
        if meta['model'] == 'mask_rcnn':
            m = torch.zeros(1, h, w, dtype=torch.uint8)
            m[:, h // 8:h // 2, w // 8:w // 2] = 1
            t['masks'] = m
        if meta['model'] == 'keypoint_rcnn':
            kp = torch.rand(1, 17, 3, generator=g)
            kp[..., 0] *= w
            kp[..., 1] *= h
            kp[..., 2] = 1
            t['keypoints'] = kp
        targets.append(t)
    return images, targets


def checksum(t, nsamples=64):
    f = t.detach().double().flatten().cpu()
    n = min(nsamples, f.numel())
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // max(n - 1, 1)
    return float(f.sum()), float((f * f).sum()), f[idx].clone()


def compare(z, name, t, rtol, what='', atol=0.0):
    """Compare tensor ``t`` with fixture entry ``name`` (full tensor or checksum form).
    Returns the achieved relative error (L2 for full tensors; max over checksum parts otherwise)."""
    t = t.detach().cpu()
    if name in z.files:
        ref = torch.from_numpy(z[name]).double()
        err = float((t.double() - ref).norm() / (ref.norm() + atol * ref.numel() ** 0.5 + 1e-30))
    else:
        s, ss, samples = checksum(t)
        assert tuple(z[name + '@shape']) == tuple(t.shape), (name, tuple(z[name + '@shape']), tuple(t.shape))
        ref_samples = torch.from_numpy(z[name + '@samples'])
        scale = float(np.sqrt(float(z[name + '@sumsq']) / t.numel())) + 1e-30   # rms of the tensor
        e_samples = float((samples - ref_samples).abs().max()) / scale
        e_ss = abs(ss - float(z[name + '@sumsq'])) / (abs(float(z[name + '@sumsq'])) + 1e-30)
        e_s = abs(s - float(z[name + '@sum'])) / (scale * t.numel() ** 0.5 * 8 + abs(float(z[name + '@sum'])))
        err = max(e_samples, e_ss, e_s)
    assert err <= rtol, '%s %s: rel err %.3e > %.1e' % (what, name, err, rtol)
    return err


def coco_eval_case_inputs(seed=51):
    """Re-create the seeded inputs of make_golden.coco_eval_case_inputs (kept in sync by test_coco_eval_cpu)."""
    g = torch.Generator().manual_seed(seed)
    dataset, preds = [], {}
    for i in range(6):
        h, w = 360 + 30 * i, 480 + 24 * i
        n = [3, 1, 5, 2, 0, 4][i]
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        wh = torch.rand(n, 2, generator=g) ** 2 * torch.tensor([w * 0.45, h * 0.45]) + 6
        boxes = torch.cat([xy, xy + wh], 1)
        labels = torch.randint(1, 4, (n,), generator=g)
        crowd = (torch.rand(n, generator=g) < 0.15).to(torch.int64)
        tgt = {'image_id': torch.tensor([200 + i]), 'boxes': boxes, 'labels': labels,
               'area': (wh[:, 0] * wh[:, 1]), 'iscrowd': crowd}
        dataset.append((torch.zeros(3, h, w), tgt))
        keep = torch.rand(n, generator=g) < 0.8
        jit = (torch.rand(n, 4, generator=g) - 0.5) * torch.cat([wh, wh], 1) * 0.22
        pb = (boxes + jit)[keep]
        pl = labels[keep].clone()
        if len(pl) > 1:
            pl[0] = 1 + (pl[0] % 3)
        fp = torch.rand(3, 2, generator=g) * torch.tensor([w * 0.7, h * 0.7])
        fpb = torch.cat([fp, fp + torch.rand(3, 2, generator=g) * 30 + 3], 1)
        pb = torch.cat([pb, pb[:1], fpb], 0)
        pl = torch.cat([pl, pl[:1], torch.randint(1, 4, (3,), generator=g)], 0)
        ps = torch.rand(len(pb), generator=g)
        preds[200 + i] = {'boxes': pb, 'labels': pl, 'scores': ps}
    return dataset, preds


def coco_eval_case_extras(dataset, preds, kind, seed=77):
    """Masks or keypoints on top of coco_eval_case_inputs (its random stream is left untouched): elliptical instance
    masks inside each box, probabilities for the predictions; 17 keypoints per instance with visibility 0 / 1 / 2
    (one ground-truth instance without any labelled keypoint), jittered copies for the predictions.  Used by
    tests/golden/make_golden.py (reference evaluator -> fixture) and by the product evaluator's tests."""
    assert kind in ('segm', 'keypoints')
    g = torch.Generator().manual_seed(seed)

    def ellipse(box, h, w, shrink=1.0):
        yy, xx = torch.meshgrid(torch.arange(h).float() + 0.5, torch.arange(w).float() + 0.5, indexing='ij')
        cx, cy = (box[0] + box[2]) / 2, (box[1] + box[3]) / 2
        rx, ry = (box[2] - box[0]) / 2 * shrink, (box[3] - box[1]) / 2 * shrink
        return ((xx - cx) / rx.clamp(min=0.5)) ** 2 + ((yy - cy) / ry.clamp(min=0.5)) ** 2 <= 1.0

    out_ds, out_preds = [], {}
    for img, tgt in dataset:
        h, w = img.shape[-2:]
        t = {k: v.clone() for k, v in tgt.items()}
        n = len(t['boxes'])
        if kind == 'keypoints' and n == 0:
            continue            # the reference's convert_to_coco_api cannot reshape an empty keypoint tensor (:175)
        if kind == 'segm':
            t['masks'] = torch.stack([ellipse(b, h, w) for b in t['boxes']]).to(torch.uint8) if n else \
                torch.zeros(0, h, w, dtype=torch.uint8)
            t['area'] = t['masks'].flatten(1).sum(1).float() if n else t['area']
        else:
            kp = torch.rand(n, 17, 3, generator=g)
            wh = (t['boxes'][:, 2:] - t['boxes'][:, :2])[:, None]
            kp[..., :2] = t['boxes'][:, None, :2] + kp[..., :2] * wh
            kp[..., 2] = torch.randint(0, 3, (n, 17), generator=g).float()
            if n > 2:
                kp[2, :, 2] = 0                                          # an instance with num_keypoints == 0
            t['keypoints'] = kp
        out_ds.append((img, t))
        p = {k: v.clone() for k, v in preds[int(tgt['image_id'])].items()}
        m = len(p['boxes'])
        if kind == 'segm':
            probs = torch.stack([ellipse(b, h, w, 0.9).float() * 0.8 + 0.1 for b in p['boxes']])[:, None]
            p['masks'] = (probs + (torch.rand(probs.shape, generator=g) - 0.5) * 0.3).clamp(0, 1)
        else:
            wh = (p['boxes'][:, 2:] - p['boxes'][:, :2])[:, None]
            kp = torch.rand(m, 17, 3, generator=g)
            kp[..., :2] = p['boxes'][:, None, :2] + kp[..., :2] * wh
            # detections that copy a ground-truth instance: its keypoints, jittered by a few percent of the box
            for j in range(min(m, n)):
                if float((p['boxes'][j] - t['boxes'][j]).abs().max()) < 0.25 * float(wh[j].max()):
                    kp[j, :, :2] = t['keypoints'][j, :, :2] + (torch.rand(17, 2, generator=g) - 0.5) * 0.08 * wh[j]
            kp[..., 2] = 1
            p['keypoints'] = kp
        out_preds[int(tgt['image_id'])] = p
    return out_ds, out_preds
