#!/usr/bin/env python
"""Neural-filter training step on one MI355X: images/s of
transform -> stem -> Ext4ResNet -> (as written) layer1 in train mode -> cross entropy -> backward -> SGD.

    python tools/bench_filter.py [--batch 2] [--steps 20] [--warmup 5] [--cpu_steps 2]

Prints one JSON line; `cpu_baseline` is the oracle (oracle.hnd_oracle.FilterOracle) on the host cores.
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
    ap.add_argument('--batch', type=int, default=2, help='config/ext: train.batch_size 2')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1333)
    ap.add_argument('--cpu_steps', type=int, default=2)
    ap.add_argument('--cpu_threads', type=int, default=32)
    ap.add_argument('--graph', action='store_true', help='replay the step as a captured hipGraph (graph.GraphedStep)')
    args = ap.parse_args()
    from hnd_ghnd_object_detectors_amd import ext_runner
    from hnd_ghnd_object_detectors_amd.configs import make_ext_config
    from hnd_ghnd_object_detectors_amd.models import get_model
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util, module_util
    from hnd_ghnd_object_detectors_amd.utils import data_util
    dev = torch.device('cuda:0')
    cfg = make_ext_config(3, batch_size=args.batch, pretrained=False, min_size=800, ckpt_root='/nonexistent')
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = get_model(cfg['model'], dev, strict=False)
    module_util.freeze_module_params(model)
    ext = model.get_ext_classifier()
    module_util.unfreeze_module_params(ext)
    model.train_ext()
    model.train()
    opt = func_util.get_optimizer(ext, 'SGD', cfg['train']['optimizer']['params'])
    loader = data_util.SyntheticDetectionLoader(1, args.batch, args.height, args.width, 'keypoint_rcnn',
                                                positive_every=2)
    images, targets = next(iter(loader))
    images = [im.to(dev) for im in images]
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]

    # image-level labels (reference src/ext_runner.py:55-56) depend on the targets only: computed once, on the host,
    # from what the model's transform leaves (so the captured step has no device -> host read in it)
    with torch.no_grad():
        probe = [dict(t) for t in targets]
        model.transform(images, probe, None)
    labels = ext_runner.convert_target2ext_targets(probe, dev)

    from hnd_ghnd_object_detectors_amd.models.ext.classifier import cross_entropy

    def body():
        tg = [dict(t) for t in targets]
        logits = model(images, tg)
        loss = cross_entropy(logits, labels)              # the product kernel (hnd_softmax_ce_rows_fwd_bwd)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.detach()

    if args.graph:
        from hnd_ghnd_object_detectors_amd.graph import GraphedStep
        graphed = GraphedStep(body, key=lambda: tuple(g['lr'] for g in opt.param_groups), warmup=3)

        def step():
            return float(graphed())
    else:
        def step():
            return float(body())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out = {'metric': 'neural-filter train-step images/sec at 3x%dx%d' % (args.height, args.width),
           'value': round(args.batch / dt, 2), 'unit': 'img/s', 'ms_per_step': round(dt * 1e3, 3),
           'batch': args.batch, 'steps': args.steps, 'dtype': 'f32', 'last_loss': last,
           'hipgraph': bool(args.graph)}
    if args.cpu_steps > 0:
        from oracle import hnd_oracle as O
        torch.set_num_threads(args.cpu_threads)
        t_sd = O.init_teacher_state(1, 'keypoint_rcnn', num_classes=2)
        orc = O.FilterOracle(O.init_student_state(t_sd, 2), O.init_ext_state(3))
        ims = [im.cpu() for im in images]
        tgs = [{k: v.cpu() for k, v in t.items()} for t in targets]
        orc.step(ims, tgs)
        t0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            orc.step(ims, tgs)
        cdt = (time.perf_counter() - t0) / args.cpu_steps
        out['cpu_baseline'] = {'value': round(args.batch / cdt, 3), 'unit': 'img/s', 'cores': args.cpu_threads,
                               'kind': 'port', 'sample': '%d steps of batch %d' % (args.cpu_steps, args.batch)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
