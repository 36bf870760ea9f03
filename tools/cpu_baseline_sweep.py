#!/usr/bin/env python
"""Thread sweep of bench.py's `cpu_baseline` leg (the CPU oracle, test infrastructure) on this host: 1 warm-up + 2 timed
GHND steps of batch 2 at 3x800x1333 per thread count.  BASELINE.md section 4 asks for the best the host does; bench.py
reads the fastest count from the JSON this writes.

    python tools/cpu_baseline_sweep.py --threads 32,64,128 --out profiles/r05_cpu_baseline_sweep.json
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
    ap.add_argument('--threads', default='8,12,16,24,32')
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_cpu_baseline_sweep.json'))
    args = ap.parse_args()
    from oracle import hnd_oracle as O
    from hnd_ghnd_object_detectors_amd.configs import make_config
    from hnd_ghnd_object_detectors_amd.synthetic import build_distillation_pair
    config = make_config('faster_rcnn', 'ghnd', 3, batch_size=args.batch, pretrained=False, ckpt_root='/nonexistent')
    with contextlib.redirect_stdout(sys.stderr):
        teacher, student = build_distillation_pair(config, torch.device('cpu'), seed=0)
    t_sd = {k: v.detach().clone() for k, v in teacher.state_dict().items()}
    s_sd = {k: v.detach().clone() for k, v in student.state_dict().items()}
    terms = {k: v['factor'] for k, v in config['train']['criterion']['terms'].items()}
    g = torch.Generator().manual_seed(1234)
    images = [torch.rand(3, 800, 1333, generator=g) for _ in range(args.batch)]
    runs = []
    for n in (int(v) for v in args.threads.split(',')):
        if n > (os.cpu_count() or 1):
            print('skip %d threads: host has %d logical CPUs' % (n, os.cpu_count()), flush=True)
            continue
        torch.set_num_threads(n)
        orc = O.DistillOracle(dict(t_sd), {k: v.clone() for k, v in s_sd.items()}, terms=terms, min_size=(800,),
                              max_size=1333)
        orc.step(images)
        t0 = time.time()
        for _ in range(args.steps):
            orc.step(images)
        dt = time.time() - t0
        runs.append({'threads': n, 'img_s': round(args.batch * args.steps / dt, 4), 's_per_step': round(dt / args.steps, 3)})
        print(runs[-1], flush=True)
    out = {'host_logical_cpus': os.cpu_count(), 'torch': torch.__version__, 'batch': args.batch,
           'steps': '1 warm-up + %d timed' % args.steps, 'runs': runs}
    with open(args.out, 'w') as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
