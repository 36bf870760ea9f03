#!/usr/bin/env python
"""Validation-path throughput on one MI355X: images/s of the eval-mode detector as
src/utils/main_util.py:75-113 drives it -- batch-1 images, transform -> backbone + FPN -> RPN -> RoI heads (box, + mask
/ keypoint branches) -> postprocess, detections copied to the host like the reference loop does (masks as the
run lengths of the thresholded bits, which is what the evaluator keeps).

    python tools/bench_eval.py [--model faster_rcnn|mask_rcnn|keypoint_rcnn] [--student] [--images 20] [--cpu_images 2]

Prints one JSON line; `cpu_baseline` is the oracle detector (oracle.hnd_oracle.DetectOracle) on the host cores.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='faster_rcnn')
    ap.add_argument('--student', action='store_true', help='the bottleneck-injected student instead of the teacher')
    ap.add_argument('--images', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1333)
    ap.add_argument('--cpu_images', type=int, default=2)
    ap.add_argument('--cpu_threads', type=int, default=32)
    args = ap.parse_args()
    from hnd_ghnd_object_detectors_amd import engine as E
    from hnd_ghnd_object_detectors_amd.configs import make_config
    from hnd_ghnd_object_detectors_amd.synthetic import build_distillation_pair
    from hnd_ghnd_object_detectors_amd.utils import mask_util
    from oracle import hnd_oracle as O
    dev = torch.device('cuda:0')
    config = make_config(args.model, 'ghnd', 3, pretrained=False, ckpt_root='/nonexistent')
    with contextlib.redirect_stdout(sys.stderr):
        teacher, student = build_distillation_pair(config, dev, seed=0)
    model = student if args.student else teacher
    # seeded heads give near-uniform scores; the fixtures' head scaling makes NMS / thresholds / top-k bite
    sd = O.scale_detector_heads({k: v.detach().cpu() for k, v in model.state_dict().items()})
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    model.eval()
    model.distill_backbone_only = False
    if args.student:
        model.backbone.body.layer1.use_bottleneck_transformer = False
    g = torch.Generator().manual_seed(4321)
    images = [torch.rand(3, args.height, args.width, generator=g) for _ in range(4)]
    dimgs = [im.to(dev) for im in images]

    def one(i):
        out = model([dimgs[i % len(dimgs)]])
        res = [{k: (v if k == 'masks' else v.cpu()) for k, v in d.items()} for d in out]
        for d in res:                       # what CocoEvaluator.prepare_for_coco_segmentation does with the masks
            if 'masks' in d:
                d['masks'] = mask_util.encode_probs(d['masks'][:, 0], 0.5)
        return res

    with torch.no_grad():
        for i in range(args.warmup):
            one(i)
        torch.cuda.synchronize()
        t0 = time.time()
        ndet = 0
        for i in range(args.images):
            ndet += len(one(i)[0]['scores'])
        torch.cuda.synchronize()
        dt = time.time() - t0
    cpu = None
    if args.cpu_images > 0:
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        det = O.DetectOracle(sd, args.model, student=args.student)
        det([images[0]])
        t0 = time.time()
        for i in range(args.cpu_images):
            det([images[(i + 1) % len(images)]])
        cdt = time.time() - t0
        cpu = {'value': round(args.cpu_images / cdt, 3), 'unit': 'img/s', 'cores': torch.get_num_threads(),
               'kind': 'port', 'sample': '%d images after 1 warm-up, oracle detector' % args.cpu_images}
    print(json.dumps({'metric': 'validation images/sec at 3x%dx%d, eval-mode %s (%s), batch 1' % (
        args.height, args.width, args.model, 'student' if args.student else 'teacher'),
        'value': round(args.images / dt, 2), 'unit': 'img/s', 'ms_per_image': round(1e3 * dt / args.images, 2),
        'detections_per_image': ndet / args.images, 'images': args.images, 'warmup': args.warmup, 'dtype': 'f32',
        'data': 'synthetic', 'cpu_baseline': cpu}))


if __name__ == '__main__':
    main()
