#!/usr/bin/env python
"""Throughput benchmark of the north-star path: GHND distillation steps, Faster R-CNN ResNet-50-FPN b3ch,
batch 16 per GPU of synthetic 3x800x1333 images (network sees 800x1344), fp32, on 1..8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # starts its own 8 ranks (child processes), one per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # or ranks started by a launcher (RANK/WORLD_SIZE in env)

With --gpus N > 1 and no RANK in the environment the process becomes a LAUNCHER: it never touches the GPU, starts N
copies of itself as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, the
reference's env:// rendezvous, src/utils/main_util.py:43-62), relays rank 0's JSON line and exits with the worst
child exit code.  It refuses (non-zero) when the node has fewer than N GPUs unless --share_device is given.

One "step" = what mimic_runner.distill_model does per batch (reference src/mimic_runner.py:48-58):
DistillationBox forward (teacher + student incl. the FPN both run, as written), zero_grad, backward,
gradient all-reduce (N > 1), Adam step, loss.item().  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, measured with HIP
events on the launch stream during one extra, untimed, single-stream step after the timed loop) and `cpu_baseline`
(the CPU oracle on this host).  The first step's loss is checked against the CPU oracle's value for the same seeded
weights and batch (tests/golden/bench_first_loss.json, written by tests/golden/make_bench_loss.py) -- outside the
timed region; a mismatch aborts the run.
"""
import argparse
import contextlib
import json
import os
import re
import sys
import time

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
    ap.add_argument('--cpu_steps', type=int, default=3, help='timed oracle steps after 1 warm-up (SURVEY.md 8d)')
    ap.add_argument('--cpu_threads', type=int, default=32)
    ap.add_argument('--dist_backend', default='nccl', help="'nccl' (= RCCL); 'gloo' only to exercise the multi-rank "
                    "code path on a single-GPU box together with --share_device")
    ap.add_argument('--share_device', action='store_true', help='testing: every rank uses cuda:0')
    ap.add_argument('--rank_timeout_s', type=int, default=1500, help='self-launched ranks (--gpus N, no RANK in the '
                    'environment): wall-clock limit after which the launcher kills its children and exits non-zero')
    ap.add_argument('--ref_1gpu_img_s', type=float, default=None, help='a --gpus 1 value of the same build on the same '
                    'node: the line then carries scaling_efficiency = value / (N * this)')
    ap.add_argument('--detail', default=None, help='write the per-launch table of the profiled step to this file')
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def visible_gpu_count():
    """GPUs this process would see, WITHOUT loading torch or touching HIP: KFD topology nodes with SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one is set.  None when the topology cannot
    be read (the caller then asks torch)."""
    try:
        top = '/sys/class/kfd/kfd/topology/nodes'
        n = 0
        for node in os.listdir(top):
            props = dict(l.split() for l in open(os.path.join(top, node, 'properties')) if len(l.split()) == 2)
            n += int(props.get('simd_count', 0)) > 0
    except (OSError, ValueError):
        return None
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        val = os.environ.get(var)
        if val is not None:
            n = min(n, len([v for v in val.split(',') if v.strip() != '']))
    return n


def launch_ranks(args):
    """--gpus N without a launcher: become one.  The parent makes no HIP call (GPUs are counted from sysfs) and execs
    nothing: the ranks are plain child processes, ALL of them watched while rank 0's output is drained -- the first
    rank to fail starts a 60 s deadline for its siblings (they would otherwise sit in a rendezvous / collective until
    the process-group timeout), and --rank_timeout_s bounds the whole run; on either the parent kills the CHILDREN it
    started (by pid) and exits non-zero."""
    import subprocess
    import threading
    have = visible_gpu_count()
    if have is None:
        import torch            # (device_count() does not initialise HIP on this image)
        have = torch.cuda.device_count()
    if have < args.gpus and not args.share_device:
        sys.stderr.write('bench.py: --gpus %d but this node exposes %d GPU(s); refusing to report a %d-GPU number '
                         '(--share_device puts every rank on cuda:0 for plumbing tests only)\n'
                         % (args.gpus, have, args.gpus))
        return 2
    if have < 1:
        sys.stderr.write('bench.py needs an MI355X: the HIP path has no CPU fallback\n')
        return 2
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HND_BENCH_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: required by RCCL on this pool
        env.setdefault('OMP_NUM_THREADS', '8')
        if args.share_device:                   # several processes on one GPU: the third (FPN) stream thrashes
            env['HND_DEFER_FPN'] = '0'
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_end = time.time() + args.rank_timeout_s
    codes, why = [None] * len(procs), None
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
                if codes[r] not in (None, 0) and why is None:
                    why = 'rank %d exited with %d' % (r, codes[r])
                    t_end = min(t_end, time.time() + float(os.environ.get('HND_BENCH_SIBLING_GRACE_S', '60')))
        if time.time() > t_end:
            why = why or ('no result after --rank_timeout_s %d' % args.rank_timeout_s)
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.kill()
                    codes[r] = pr.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=10.0)
    raw = (chunks[0] if chunks else b'').decode(errors='replace')      # rank 0 prints the JSON line; native libraries
    line = ''.join(l + '\n' for l in raw.splitlines() if l.startswith('{"metric"'))     # (gloo, RCCL debug) may add theirs
    sys.stderr.write(''.join(l + '\n' for l in raw.splitlines() if l.strip() and not l.startswith('{"metric"')))
    bad = [c for c in codes if c != 0]
    if bad:
        sys.stderr.write('bench.py: %s; rank exit codes %s -- no line reported\n' % (why, codes))
        return bad[0] if bad[0] > 0 else 1
    sys.stdout.write(line)
    sys.stdout.flush()
    return 0


def gpu_cpu_affinity(local_rank):
    """bind this rank to the CPUs of the NUMA node its GPU hangs off (KFD topology -> PCI local_cpulist), before
    torch spins its thread pools up.  Best effort: returns the description that goes into the JSON line."""
    try:
        gpus = []
        top = '/sys/class/kfd/kfd/topology/nodes'
        for node in sorted(os.listdir(top), key=int):
            props = dict(l.split() for l in open(os.path.join(top, node, 'properties')) if len(l.split()) == 2)
            if int(props.get('simd_count', 0)) > 0:
                gpus.append(props)
        props = gpus[local_rank]
        loc, dom = int(props['location_id']), int(props.get('domain', 0))
        bdf = '%04x:%02x:%02x.%d' % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        cpus = open('/sys/bus/pci/devices/%s/local_cpulist' % bdf).read().strip()
        ids = set()
        for part in cpus.split(','):
            lo, _, hi = part.partition('-')
            ids.update(range(int(lo), int(hi or lo) + 1))
        ids &= os.sched_getaffinity(0)
        if not ids:
            return 'unbound (empty local_cpulist for %s)' % bdf
        os.sched_setaffinity(0, ids)
        node = open('/sys/bus/pci/devices/%s/numa_node' % bdf).read().strip()
        return 'gpu %d @ %s -> numa node %s, cpus %s' % (local_rank, bdf, node, cpus)
    except Exception as exc:            # unknown topology layout: stay unbound rather than guess
        return 'unbound (%s: %s)' % (type(exc).__name__, exc)


def physical_cores():
    """distinct (socket, core) pairs of this host (hyper-threads share one)"""
    try:
        seen, phys = set(), None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                seen.add((phys, line.split(':')[1].strip()))
        return len(seen) or None
    except OSError:
        return None


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
            'host_logical_cpus': os.cpu_count(), 'host_physical_cores': physical_cores(),
            'sample': '%d warm-up + %d timed %s steps of batch %d at 3x%dx%d, torch %s CPU oracle (oracle/hnd_oracle.py), '
                      '%d threads' % (1, args.cpu_steps, args.method.upper(), args.cpu_batch, args.height, args.width,
                                      torch.__version__, cores)}


def first_loss_reference(args, rank):
    """the CPU oracle's losses (total per method + per term) for this run's first step (same seeded weights and
    batch), or None if the configuration is not one of the pinned ones"""
    if (args.height, args.width) != (800, 1333):
        return None
    try:
        cases = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_first_loss.json')))['cases']
    except (OSError, ValueError, KeyError):
        return None
    return cases.get('%s/batch%d/rank%d' % (args.model, args.batch, rank))


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))            # this process stays off the GPU; the ranks are its children
    run_rank(args)


def run_rank(args):
    global torch, dist
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d: the line would be mislabelled' % (world, args.gpus))
    if os.environ.get('HND_BENCH_FAIL_RANK') == str(rank):      # tests: a rank that dies before the rendezvous
        raise SystemExit(3)
    affinity = gpu_cpu_affinity(0 if args.share_device else local_rank) if world > 1 else 'unbound (single rank)'
    # dmabuf IPC (RCCL needs it on this pool), also when the ranks come from torch.distributed.run: before any HIP call
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    if not args.share_device and torch.cuda.device_count() < world:
        raise SystemExit('bench.py: %d ranks but %d visible GPU(s)' % (world, torch.cuda.device_count()))
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
    warm = main_util.warmup_lr_scheduler(optimizer, 1000, 1.0 / 1000.0)     # epoch-0 warm-up, mimic_runner.py:43-46

    g = torch.Generator().manual_seed(1234 + rank)              # per-rank seeded synthetic shard
    images = [torch.rand(3, args.height, args.width, generator=g).to(dev) for _ in range(args.batch)]
    h, w = args.height, args.width
    targets = [{'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]], device=dev),
                'labels': torch.tensor([1], device=dev)} for _ in range(args.batch)]

    term_names = list(config['train']['criterion']['terms'].keys())
    term_values = {}

    def step(sync=True, keep_terms=False):
        loss = box(images, [dict(t) for t in targets])
        if keep_terms:                                          # the fused loss launch's per-term sums (fp64)
            term_values.update(zip(term_names, (float(v) for v in loss.per_term.tolist())))
        optimizer.zero_grad()
        loss.backward()                                         # N > 1: fires the flat gradient all-reduce (RCCL)
        optimizer.step()                                        # waits for it stream-side, applies the 1/world mean
        warm.step()
        return loss.item() if sync else loss                    # MetricLogger.update -> one host sync per step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- parity gate, outside the timed region: this run's very first step against the CPU oracle -- the total AND
    # every term (layer1 is 96 % of the GHND total: a gate on the total alone could not see layer3 / layer4 at all)
    first = step(keep_terms=True)
    ref_case = first_loss_reference(args, rank)
    loss_check, bad = None, 0
    if ref_case is not None:
        ref_first = ref_case[args.method]
        rel = abs(first - ref_first) / abs(ref_first)
        terms_chk = {k: {'got': term_values[k], 'oracle': ref_case['terms'][k],
                         'rel_err': abs(term_values[k] - ref_case['terms'][k]) / abs(ref_case['terms'][k])}
                     for k in term_names}
        worst = max([rel] + [t['rel_err'] for t in terms_chk.values()])
        loss_check = {'first_step_loss': first, 'oracle': ref_first, 'rel_err': rel, 'terms': terms_chk,
                      'worst_rel_err': worst, 'tol': 1e-3, 'source': 'tests/golden/bench_first_loss.json'}
        bad = 0 if worst < 1e-3 else 1
    elif (args.height, args.width, args.model) == (800, 1333, 'faster_rcnn') and args.batch == 16:
        bad = 1                     # the benchmarked configuration must never run ungated
    if world > 1:                   # every rank learns the verdict, so nobody is left waiting in a collective
        flag = torch.tensor([bad, 0 if loss_check is None else 1], dtype=torch.int32, device=dev)
        worst_t = torch.tensor([0.0 if loss_check is None else loss_check['worst_rel_err']], dtype=torch.float64,
                               device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        dist.all_reduce(worst_t, op=dist.ReduceOp.MAX)
        any_bad, gated_ranks, worst_all = int(flag[0].item()), int(flag[1].item()), float(worst_t.item())
    else:
        any_bad, gated_ranks = bad, 0 if loss_check is None else 1
        worst_all = None if loss_check is None else loss_check['worst_rel_err']
    if any_bad:
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit('bench.py: first-step loss (total or a term) differs from the CPU oracle\'s by more than 1e-3 '
                         'on %d rank(s) of this run, or the benchmarked configuration has no pinned loss (rank %d: %r): '
                         'the HIP path is wrong at the benchmarked configuration' % (any_bad, rank, loss_check))
    if loss_check is not None:
        loss_check['ranks_gated'] = gated_ranks
        loss_check['worst_rel_err_all_ranks'] = worst_all
    last = first
    for _ in range(args.warmup - 1):
        last = step()
    if world > 1:
        student_w.timing = True
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    exchange_ms = None
    if world > 1:
        student_w.timing = False
        exchange_ms = student_w.exchange_ms()
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [round(v / args.steps * 1e3, 3) for v in t.tolist()]
        elapsed = float(t.max().item())                          # MAX over ranks
    exchange_ms_ranks = None
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = exchange_ms or 0.0
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        exchange_ms_ranks = [round(v, 4) for v in t.tolist()]
    rccl_ranks = dist.get_world_size() if (world > 1 and dist.get_backend() == 'nccl') else (1 if world == 1 else 0)
    exchange = 'none (1 rank)'
    if world > 1:
        exchange = '%s all-reduce of the flat %d-float gradient arena, %d fired in %d steps on this rank' % (
            'RCCL (torch.distributed nccl)' if dist.get_backend() == 'nccl' else dist.get_backend(),
            student.backbone.body._grad_arena.total, student_w.reductions, args.warmup + args.steps)

    # ---- untimed extras (every rank runs them: the steps contain the all-reduce)
    # host enqueue cost: wall time to ISSUE a step with no host sync in it (what one Python process per GPU pays)
    nh = 3
    fence()
    h0 = time.perf_counter()
    for _ in range(nh):
        step(sync=False)
    host_enqueue_ms = (time.perf_counter() - h0) / nh * 1e3
    fence()
    # per-launch HIP events: one extra single-stream step (kernels timed running alone)
    from hnd_ghnd_object_detectors_amd import ops as OPS
    E.PROFILE['enabled'], E.PROFILE['records'] = True, []
    OPS.HBM_PROFILE['enabled'], OPS.HBM_PROFILE['records'] = True, []
    step()
    E.PROFILE['enabled'] = OPS.HBM_PROFILE['enabled'] = False
    fence()
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel from the per-launch HIP events of the profiled step
    per, groups, hbm = {}, {}, {}

    def hbm_add(kernel, nbytes, ms):
        h_ = hbm.setdefault(kernel, {'launches': 0, 'ms': 0.0, 'bytes': 0})
        h_['launches'] += 1
        h_['ms'] += ms
        h_['bytes'] += nbytes
    for kernel, nbytes, e0, e1 in OPS.HBM_PROFILE['records']:
        hbm_add(kernel, nbytes, e0.elapsed_time(e1))
    re_3x3 = re.compile(r'^(layer\d\.\d+\.conv2|fpn\.layer\d)(\.dgrad)?(\.wino_(in|out))?$')
    re_head = re.compile(r'^(layer1\.conv\d)(\.dgrad|\.wgrad)?(\.wino_(in|out|dy))?$')
    for tag, launch, e0, e1 in E.PROFILE['records']:
        ms = e0.elapsed_time(e1)
        if getattr(launch, 'hbm_bytes', 0):              # Winograd transforms: HBM-bound plan entries
            hbm_add(launch.kernel, launch.hbm_bytes, ms)
        if launch.flops:
            d = per.setdefault(launch.variant, {'ms': 0.0, 'flop': 0.0, 'n': 0})
            d['ms'] += ms
            d['flop'] += launch.flops
            d['n'] += 1
        m3, mh = re_3x3.match(tag), re_head.match(tag)
        if m3:
            gkey = ('conv3x3', 'all')
        elif mh:
            gkey = ('head2x2', mh.group(1)[len('layer1.'):] + (mh.group(2) or '.fwd'))
        else:
            continue
        g_ = groups.setdefault(gkey, {'ms': 0.0, 'alg': 0.0})
        g_['ms'] += ms
        g_['alg'] += launch.alg_flops
    kernels = {k: {'launches': v['n'], 'ms': round(v['ms'], 3), 'tflops': round(v['flop'] / v['ms'] / 1e9, 2)}
               for k, v in per.items() if v['ms'] > 0}
    dom = max(per, key=lambda k: per[k]['ms'])
    ach = per[dom]['flop'] / per[dom]['ms'] / 1e9
    roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                'traffic_measured_in_run': False,
                'launches_per_step': per[dom]['n'], 'avg_launch_ms': round(per[dom]['ms'] / per[dom]['n'], 4),
                'algorithmic_gflop_per_launch': round(per[dom]['flop'] / per[dom]['n'] / 1e9, 3),
                'flops_basis': ('multiplies the launches execute: implicit-GEMM / B-resident GEMM convs 2*M*Cout*K; the '
                                'Winograd component GEMMs 2*ncomp*tiles*Cin*Cout (F(4x4,3x3): 36 products per 16 outputs, '
                                'F(6x6,3x3): 64 per 36, F(4x4,2x2): 25 per 16 -- 4x / 5.06x / 2.56x fewer than the direct '
                                'convolutions they replace)') if E.WINOGRAD else 'implicit-GEMM convs 2*M*Cout*K',
                'per_kernel_frac': {k: round(v['flop'] / v['ms'] / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4)
                                    for k, v in per.items() if v['ms'] > 0}}
    run_cfg = {'batch': args.batch, 'method': args.method, 'model': args.model, 'winograd': E.WINOGRAD,
               'winograd6': bool(E.WINOGRAD6), 'bres': os.environ.get('HND_BRES', '512'),
               'no_fpn': bool(args.no_fpn), 'height': args.height, 'width': args.width}
    try:        # HBM bytes per launch of the dominant kernel from the newest committed PMC pass OF THIS CONFIGURATION
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
            prof = json.load(open(tf))
            if prof.get('config') == run_cfg and dom in prof['kernels']:
                roofline['traffic'] = prof['kernels'][dom]['traffic_bytes_per_launch']
                roofline['traffic_source'] = os.path.basename(tf) + ' (separate rocprofv3 --pmc passes of this config)'
                break
        else:
            roofline['traffic_source'] = 'no committed PMC profile matches this configuration'
    except Exception as exc:        # a malformed profile file must not lose the measurement
        roofline['traffic_source'] = 'unreadable profile: %s' % exc
    # "3x3-conv roofline fraction" = algorithmic 2*MAC of the 3x3 convs (frozen conv2 fwd + dgrad, FPN
    # output convs) / their summed kernel time INCLUDING the Winograd transforms / fp32 MFMA peak; and the 2x2 head
    g3 = groups.get(('conv3x3', 'all'))
    conv3x3 = None
    if g3 and g3['ms'] > 0:
        tf3 = g3['alg'] / g3['ms'] / 1e9
        conv3x3 = {'algorithmic_gflop': round(g3['alg'] / 1e9, 1), 'ms': round(g3['ms'], 3),
                   'algorithmic_tflops': round(tf3, 2),
                   'algorithmic_speed_vs_direct_at_peak': round(tf3 / FP32_MFMA_PEAK_TFLOPS, 4),
                   'note': 'direct-convolution 2*MAC / kernel time incl. the Winograd transforms; NOT a roofline '
                           'fraction (Winograd executes 4x fewer multiplies): > 1 means faster than a direct conv at peak'}
    head2x2 = {k[1]: {'algorithmic_gflop': round(v['alg'] / 1e9, 2), 'ms': round(v['ms'], 3),
                      'algorithmic_tflops': round(v['alg'] / v['ms'] / 1e9, 2),
                      'algorithmic_speed_vs_direct_at_peak': round(v['alg'] / v['ms'] / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4)}
               for k, v in sorted(groups.items()) if k[0] == 'head2x2' and v['ms'] > 0}
    conv_ms = sum(v['ms'] for v in per.values())
    exec_gflop = sum(v['flop'] for v in per.values()) / 1e9          # multiplies the MFMA launches execute (x2)
    # HBM side: algorithmic bytes / in-run HIP-event time of the bandwidth-bound kernels, against the 8 TB/s spec
    # and the ~6.3 TB/s a streaming kernel reaches on this part (MI355X_MICROARCH.md)
    hbm_roofline = {k: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'gbytes': round(v['bytes'] / 1e9, 3),
                        'tb_per_s': round(v['bytes'] / v['ms'] / 1e9, 3),
                        'frac_of_8.0': round(v['bytes'] / v['ms'] / 1e9 / 8.0, 3),
                        'frac_of_6.3': round(v['bytes'] / v['ms'] / 1e9 / 6.3, 3)}
                    for k, v in sorted(hbm.items(), key=lambda kv: -kv[1]['ms']) if v['ms'] > 0}
    hbm_ms = sum(v['ms'] for v in hbm.values())
    E.PROFILE['records'] = [r for r in E.PROFILE['records'] if r[1].flops]      # --detail lists the MFMA launches
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
                   'conv3x3': ('Winograd F(%dx%d,3x3)%s for stride-1 3x3 convs with >=%d channels, implicit GEMM elsewhere'
                               % (E.WINOGRAD, E.WINOGRAD, ' / F(6x6,3x3) on maps of >= %d 6x6 tiles' % E.WINOGRAD6_MIN_TILES
                                  if E.WINOGRAD6 and E.WINOGRAD == 4 else '', 128 if E.WINOGRAD == 4 else 256))
                              if E.WINOGRAD else 'implicit GEMM'},
        'roofline': roofline,
        'conv3x3_roofline': conv3x3,
        'head2x2_roofline': head2x2,
        'loss_check': loss_check,
        'ranks': {'world': world, 'rccl_ranks': rccl_ranks, 'backend': dist.get_backend() if world > 1 else None,
                  'ms_per_step_per_rank': rank_ms,
                  'ms_per_step_spread': round((max(rank_ms) - min(rank_ms)) / max(rank_ms), 4),
                  'exchange': exchange,
                  # HIP events on the compute stream of every rank, timed steps only: last gradient kernel done ->
                  # reduced arena visible to the optimizer launch (the exposed part of the all-reduce); max over ranks
                  'exchange_ms_per_step': None if exchange_ms_ranks is None else max(exchange_ms_ranks),
                  'exchange_ms_per_step_per_rank': exchange_ms_ranks, 'rank0_affinity': affinity,
                  'launched_by': 'bench.py' if os.environ.get('HND_BENCH_LAUNCHED') else
                                 ('external launcher' if world > 1 else 'single process')},
        # the driver computes scaling efficiency itself from the per-N lines; this is only filled when the caller hands
        # over a --gpus 1 figure of the same build on the same node
        'scaling_efficiency': (round(value / (world * args.ref_1gpu_img_s), 4) if args.ref_1gpu_img_s else None),
        'run_cfg': run_cfg,
        'host_enqueue_ms_per_step': round(host_enqueue_ms, 3),
        # algorithmic: direct-convolution flops the step stands for / step time -- NOT a roofline fraction (Winograd
        # executes fewer multiplies); the executed basis follows
        'step_conv_tflops': round(gflop_img * args.batch / (ms_per_step / 1e3) / 1e3, 2),
        'executed_gflop_per_step': round(exec_gflop, 1),
        'executed_tflops_in_mfma_kernels': round(exec_gflop / conv_ms, 2) if conv_ms else None,
        'executed_frac_of_mfma_peak': round(exec_gflop / conv_ms / FP32_MFMA_PEAK_TFLOPS, 4) if conv_ms else None,
        'conv_kernel_ms_per_step': round(conv_ms, 2),
        'hbm_kernel_ms_per_step': round(hbm_ms, 2),
        'hbm_roofline': hbm_roofline,
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
