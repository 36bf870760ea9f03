#!/usr/bin/env python
"""Throughput benchmark of the north-star path: GHND distillation steps, Faster R-CNN ResNet-50-FPN b3ch,
batch 16 per GPU of synthetic 3x800x1333 images (network sees 800x1344), fp32, on 1..8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = what mimic_runner.distill_model does per batch (reference src/mimic_runner.py:48-58):
DistillationBox forward (teacher + student incl. the FPN both run, as written), zero_grad, backward,
gradient all-reduce (N > 1), Adam step, loss.item().  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, measured with HIP
events on the launch stream during the last timed step) and `cpu_baseline` (the CPU oracle on this host).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, Matrix cores: v_mfma_f32_32x32x2_f32
GFLOP_PER_IMAGE = {'ghnd': 945.43, 'hnd': 803.95}      # algorithmic conv FLOPs at 800x1344


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--method', default='ghnd', choices=['ghnd', 'hnd'])
    ap.add_argument('--model', default='faster_rcnn')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1333)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_fpn', action='store_true', help='elide the loss-dead FPN (reported separately, never default)')
    ap.add_argument('--cpu_batch', type=int, default=2)
    ap.add_argument('--cpu_steps', type=int, default=1)
    ap.add_argument('--cpu_threads', type=int, default=32)
    ap.add_argument('--dist_backend', default='nccl', help="'nccl' (= RCCL); 'gloo' only to exercise the multi-rank "
                    "code path on a single-GPU box together with --share_device")
    ap.add_argument('--share_device', action='store_true', help='testing: every rank uses cuda:0')
    ap.add_argument('--detail', default=None, help='write the per-launch table of the profiled step to this file')
    return ap.parse_args()


def cpu_baseline(teacher, student, args, terms):
    """The CPU oracle (restatement of the reference path, pinned to reference-generated fixtures) timed on this
    host's cores on a bounded sample: batch `cpu_batch`, 1 warm-up + `cpu_steps` timed steps."""
    from oracle import hnd_oracle as O          # checker / baseline only -- never on the product path
    # torch/oneDNN degrades badly when oversubscribed on the 256-thread GPU host: use at most 32 threads
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    t_sd = {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()}
    s_sd = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    orc = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=(800,), max_size=1333)
    g = torch.Generator().manual_seed(1234)
    images = [torch.rand(3, args.height, args.width, generator=g) for _ in range(args.cpu_batch)]
    orc.step(images)
    t0 = time.time()
    for _ in range(args.cpu_steps):
        orc.step(images)
    dt = time.time() - t0
    return {'value': round(args.cpu_batch * args.cpu_steps / dt, 4), 'unit': 'img/s', 'cores': cores, 'kind': 'port',
            'sample': '%d warm-up + %d timed %s steps of batch %d at 3x%dx%d, torch %s CPU oracle (oracle/hnd_oracle.py), '
                      '%d threads' % (1, args.cpu_steps, args.method.upper(), args.cpu_batch, args.height, args.width,
                                      torch.__version__, cores)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        kw = {'device_id': dev} if args.dist_backend == 'nccl' else {}
        dist.init_process_group(args.dist_backend, init_method='env://', **kw)        # 'nccl' is RCCL on ROCm

    from hnd_ghnd_object_detectors_amd import engine as E
    from hnd_ghnd_object_detectors_amd.configs import make_config
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.parallel import DistributedStudent
    from hnd_ghnd_object_detectors_amd.synthetic import build_distillation_pair
    from hnd_ghnd_object_detectors_amd.utils import main_util

    config = make_config(args.model, args.method, 3, batch_size=args.batch, pretrained=False,
                         ckpt_root='/nonexistent')
    _print = print
    with contextlib.redirect_stdout(sys.stderr):        # keep stdout to the single JSON line
        teacher, student = build_distillation_pair(config, dev, seed=0)
    if args.no_fpn:
        teacher.backbone.run_fpn = student.backbone.run_fpn = False
    student_w = DistributedStudent(student) if world > 1 else student
    box = DistillationBox(teacher, student_w, config['train']['criterion'])
    optimizer = func_util.get_optimizer(student, 'Adam', config['train']['optimizer']['params'])
    if world > 1:
        student_w.attach_optimizer(optimizer)
    warm = main_util.warmup_lr_scheduler(optimizer, 1000, 1.0 / 1000.0)     # epoch-0 warm-up, mimic_runner.py:43-46

    g = torch.Generator().manual_seed(1234 + rank)              # per-rank seeded synthetic shard
    images = [torch.rand(3, args.height, args.width, generator=g).to(dev) for _ in range(args.batch)]
    h, w = args.height, args.width
    targets = [{'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]], device=dev),
                'labels': torch.tensor([1], device=dev)} for _ in range(args.batch)]

    def step():
        loss = box(images, [dict(t) for t in targets])
        optimizer.zero_grad()
        loss.backward()
        if world > 1:
            student_w.reduce_gradients()
        optimizer.step()
        warm.step()
        return loss.item()                                      # MetricLogger.update -> one host sync per step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        last = step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            E.PROFILE['enabled'], E.PROFILE['records'] = True, []
        last = step()
    E.PROFILE['enabled'] = False
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel from the per-launch HIP events of the last timed step
    per = {}
    for tag, launch, e0, e1 in E.PROFILE['records']:
        d = per.setdefault(launch.variant, {'ms': 0.0, 'flop': 0.0, 'n': 0})
        d['ms'] += e0.elapsed_time(e1)
        d['flop'] += launch.flops
        d['n'] += 1
    kernels = {k: {'launches': v['n'], 'ms': round(v['ms'], 3), 'tflops': round(v['flop'] / v['ms'] / 1e9, 2)}
               for k, v in per.items() if v['ms'] > 0}
    dom = max(per, key=lambda k: per[k]['ms'])
    ach = per[dom]['flop'] / per[dom]['ms'] / 1e9
    roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                'launches_per_step': per[dom]['n'], 'avg_launch_ms': round(per[dom]['ms'] / per[dom]['n'], 4),
                'algorithmic_gflop_per_launch': round(per[dom]['flop'] / per[dom]['n'] / 1e9, 3),
                'flops_basis': ('multiplies the launches execute: implicit-GEMM convs 2*M*Cout*K; the Winograd '
                                'F(%dx%d,3x3) GEMMs 2*%d*tiles*Cin*Cout (%.2fx fewer than the direct 3x3 they replace)'
                                % (E.WINOGRAD, E.WINOGRAD, (E.WINOGRAD + 2) ** 2,
                                   9.0 * E.WINOGRAD ** 2 / (E.WINOGRAD + 2) ** 2))
                               if E.WINOGRAD else 'implicit-GEMM convs 2*M*Cout*K'}
    try:        # HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/rNN_traffic.json)
        import glob
        tf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))[-1]
        roofline['traffic'] = json.load(open(tf))['kernels'][dom]['traffic_bytes_per_launch']
        roofline['traffic_source'] = os.path.basename(tf)
    except Exception:
        pass
    conv_ms = sum(v['ms'] for v in per.values())
    if args.detail:
        agg = {}
        for tag, launch, e0, e1 in E.PROFILE['records']:
            a = agg.setdefault((tag, launch.variant), [0.0, 0.0, 0, ''])
            a[0] += e0.elapsed_time(e1)
            a[1] += launch.flops
            a[2] += 1
            d = launch.desc
            if launch.variant.startswith('igemm'):           # GEMM extent and grid fill: tiles / resident slots
                bm, bn = (int(v) for v in launch.variant.split('_')[-1].split('x'))
                m = d.n * d.oh * d.ow
                tiles = -(-m // bm) * -(-d.cout // bn)
                slots = 256 * (3 if (bm, bn) == (128, 128) else 4)
                a[3] = 'M=%-8d N=%-5d K=%-5d tiles=%-6d rounds=%.2f' % (m, d.cout, d.kdim, tiles, tiles / slots)
        with open(args.detail, 'w') as fp:
            fp.write('%-34s %-14s %3s %9s %9s %8s  %s\n' % ('launch', 'kernel', 'n', 'ms', 'GFLOP', 'TFLOP/s', 'shape'))
            for (tag, var), (ms, fl, n, shape) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                fp.write('%-34s %-14s %3d %9.3f %9.2f %8.2f  %s\n' % (tag, var, n, ms, fl / 1e9, fl / ms / 1e9, shape))
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    gflop_img = GFLOP_PER_IMAGE[args.method] - (243.6 if args.no_fpn else 0.0)
    out = {
        'metric': 'distill-step images/sec at 3x800x1333, GHND Faster R-CNN b3ch',
        'value': round(value, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s %s ResNet50-FPN b3ch distill step, batch %d/GPU, 3x%dx%d, fp32, Adam, FPN %s'
                               % (args.method.upper(), args.model, args.batch, args.height, args.width,
                                  'elided' if args.no_fpn else 'executed (as written)'),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                   'weights': 'seeded random init (no network for COCO weights)',
                   'conv3x3': ('Winograd F(%dx%d,3x3) for stride-1 3x3 convs with >=%d channels, implicit GEMM elsewhere'
                               % (E.WINOGRAD, E.WINOGRAD, 128 if E.WINOGRAD == 4 else 256))
                              if E.WINOGRAD else 'implicit GEMM'},
        'roofline': roofline,
        'step_conv_tflops': round(gflop_img * args.batch / (ms_per_step / 1e3) / 1e3, 2),
        'conv_kernel_ms_per_step': round(conv_ms, 2),
        'kernels': kernels,
        'last_loss': last,
        'max_mem_gb': round(torch.cuda.max_memory_allocated() / 1e9, 2),
    }
    if world == 1 and not args.no_cpu_baseline:
        terms = {k: v['factor'] for k, v in config['train']['criterion']['terms'].items()}
        out['cpu_baseline'] = cpu_baseline(teacher, student, args, terms)
    else:
        out['cpu_baseline'] = None
    _print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
