"""bench.py as the driver runs it: `python bench.py --gpus N` with NO launcher environment must start its own N
ranks (row e2 of VERDICT r2), gate every rank on the pinned per-term losses and label the line with the real rank
count; asking for more GPUs than the node has must fail, not measure one GPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH_ENV = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE',
              'HND_BENCH_LAUNCHED', 'GROUP_RANK', 'TORCHELASTIC_RUN_ID')


def _run(argv, timeout=1500, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_launcher_refuses_more_gpus_than_the_node_has():
    """runs everywhere (no GPU call in the launcher): 64 GPUs exist on no node"""
    r = _run(['--gpus', '64', '--steps', '1', '--warmup', '1'], timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == b''
    assert b'refusing' in r.stderr or b'needs an MI355X' in r.stderr


def test_mislabelled_world_size_is_refused():
    """a launcher that started 1 rank for --gpus 8 must not produce an `n_gpus: 1` line (bench.py r2 :112-116 did)"""
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '1'], timeout=300,
             extra_env={'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and r.stdout.strip() == b'' and b'WORLD_SIZE=1 but --gpus 8' in r.stderr


def test_affinity_probe_never_raises():
    sys.path.insert(0, ROOT)
    import bench
    before = os.sched_getaffinity(0)
    try:
        msg = bench.gpu_cpu_affinity(0)
        assert isinstance(msg, str) and msg
        assert os.sched_getaffinity(0)            # never an empty set
    finally:
        os.sched_setaffinity(0, before)


def test_print_topology_touches_no_gpu_and_needs_no_torch():
    """--print_topology lists KFD GPU -> PCI BDF -> NUMA node -> cpulist from sysfs alone (the first thing to run on a
    new 8-GPU node: it cannot hang on a sick device).  Here (no /sys/class/kfd) it says so and exits 1."""
    code = ('import sys, runpy; sys.argv = ["bench.py", "--print_topology"]\n'
            'try:\n    runpy.run_path(%r, run_name="__main__")\n'
            'except SystemExit as e:\n    rc = e.code\n'
            'assert "torch" not in sys.modules, "torch was imported"\nsys.exit(rc)' % os.path.join(ROOT, 'bench.py'))
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    out = r.stdout.decode()
    assert r.returncode in (0, 1), r.stderr.decode()[-2000:]
    assert ('local_rank 0 -> kfd node' in out) if r.returncode == 0 else ('no readable KFD topology' in out)


@pytest.mark.gpu
def test_bench_five_ranks_on_a_shared_device_with_two_cpus_each():
    """the first-N=8-run rehearsal on one GPU -- FIVE children through the launcher (the
    pool's process guard allows 6 processes on a card: this pytest process + 5 ranks; gloo, batch 2, an ungated
    geometry), per-rank times, the exchange timed on every rank, every child reaped -- with the host budget of an 8-rank
    run on a 16-CPU container emulated: 2 usable CPUs per rank"""
    r = _run(['--gpus', '5', '--share_device', '--dist_backend', 'gloo', '--batch', '2', '--steps', '4', '--warmup', '2',
              '--no_cpu_baseline', '--no_runner'], extra_env={'HND_BENCH_USABLE_CPUS': '10'})
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    rk = out['ranks']
    assert out['n_gpus'] == 5 and out['config']['global_batch'] == 10 and rk['world'] == 5
    # 10 usable CPUs for 5 local ranks (= 16 for 8) = 2 per rank < 3 -> the loader hands out a rotating pool of
    # pre-generated pinned batches (still uploaded every step), generator threads capped at the rank's share, and the
    # step never waits for its data
    up = out['upload']
    assert up['host_usable_cpus'] == 10 and up['local_world'] == 5 and up['cpus_per_rank'] == 2.0
    assert up['pool_batches'] == 4 and up['mode'].startswith('rotating pool of 4') and up['loader_workers'] == 1
    assert len(up['data_wait_ms_per_step_per_rank']) == 5 and up['data_wait_ms_per_step'] < 1.0, up
    assert up['mbytes_per_step'] > 0
    print('[bench --gpus 5, 2 usable CPUs per rank] upload mode: %s; data_wait_ms_per_step %.3f (max over ranks), loader CPU '
          '%.1f ms / batch' % (up['mode'][:48], up['data_wait_ms_per_step'], up['loader_cpu_ms_per_batch']))
    assert rk['launched_by'] == 'bench.py' and rk['backend'] == 'gloo'
    assert len(rk['ms_per_step_per_rank']) == 5 and len(rk['exchange_ms_per_step_per_rank']) == 5
    assert all(v > 0 for v in rk['ms_per_step_per_rank']) and 0 <= rk['ms_per_step_spread'] < 1
    assert out['upload']['batches'] >= 3
    print('\n[bench --gpus 5, shared device, batch 2] %.1f img/s, per-rank ms %s, spread %.3f, exchange %.3f ms/step (gloo)'
          % (out['value'], rk['ms_per_step_per_rank'], rk['ms_per_step_spread'], rk['exchange_ms_per_step']))


@pytest.mark.gpu
def test_bench_launches_its_own_two_ranks_and_gates_both():
    r = _run(['--gpus', '2', '--share_device', '--dist_backend', 'gloo', '--steps', '2', '--warmup', '1',
              '--no_cpu_baseline'])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 32 and out['config']['parallelism'] == 'dp2'
    assert out['ranks']['world'] == 2 and out['ranks']['launched_by'] == 'bench.py'
    assert out['ranks']['backend'] == 'gloo' and out['ranks']['rccl_ranks'] == 0
    assert len(out['ranks']['ms_per_step_per_rank']) == 2
    lc = out['loss_check']
    assert lc['ranks_gated'] == 2 and lc['worst_rel_err_all_ranks'] < 1e-3
    assert sorted(lc['terms']) == ['layer1', 'layer2', 'layer3', 'layer4']
    assert all(t['rel_err'] < 1e-3 for t in lc['terms'].values())
    print('\n[bench --gpus 2, shared device] %.1f img/s, per-rank ms %s, worst first-step rel err %.1e'
          % (out['value'], out['ranks']['ms_per_step_per_rank'], lc['worst_rel_err_all_ranks']))


@pytest.mark.gpu
def test_bench_refuses_eight_gpus_on_a_smaller_node():
    if torch.cuda.device_count() >= 8:
        pytest.skip('this node has 8 GPUs')
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '1'], timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == b''


@pytest.mark.gpu
def test_bench_four_ranks_on_a_shared_device_report_the_exchange():
    """the 4-rank line (plumbing: one GPU shared, gloo, batch 2 -- an ungated geometry) carries the
    event-timed exchange and the per-rank spread the first real N = 8 run will be read by"""
    r = _run(['--gpus', '4', '--share_device', '--dist_backend', 'gloo', '--batch', '2', '--steps', '2', '--warmup', '1',
              '--no_cpu_baseline', '--no_runner', '--ref_1gpu_img_s', '100'])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    rk = out['ranks']
    assert out['n_gpus'] == 4 and out['config']['global_batch'] == 8 and rk['world'] == 4
    assert len(rk['ms_per_step_per_rank']) == 4 and len(rk['exchange_ms_per_step_per_rank']) == 4
    assert rk['exchange_ms_per_step'] is not None and rk['exchange_ms_per_step'] > 0
    assert 0 <= rk['ms_per_step_spread'] < 1
    assert abs(out['scaling_efficiency'] - out['value'] / 400.0) < 1e-3
    print('\n[bench --gpus 4, shared device, batch 2] %.1f img/s, exchange %.3f ms/step (gloo), spread %.3f'
          % (out['value'], rk['exchange_ms_per_step'], rk['ms_per_step_spread']))


@pytest.mark.gpu
def test_launcher_stops_the_siblings_of_a_rank_that_died_at_start_up():
    """rank 1 dies before the rendezvous; rank 0 would wait in it for the process-group timeout.  The
    launcher watches every child while it drains rank 0's pipe, gives the survivors a grace period, kills them by pid
    and exits non-zero with no line."""
    import time
    t0 = time.time()
    r = _run(['--gpus', '2', '--share_device', '--dist_backend', 'gloo', '--batch', '2', '--steps', '1', '--warmup', '1',
              '--no_cpu_baseline'], timeout=600, extra_env={'HND_BENCH_FAIL_RANK': '1', 'HND_BENCH_SIBLING_GRACE_S': '5'})
    assert r.returncode != 0 and r.stdout.strip() == b''
    assert b'rank 1 exited with 3' in r.stderr and time.time() - t0 < 240


@pytest.mark.gpu
def test_launcher_rank_timeout_kills_the_children():
    r = _run(['--gpus', '2', '--share_device', '--dist_backend', 'gloo', '--steps', '50', '--warmup', '1',
              '--no_cpu_baseline', '--rank_timeout_s', '8'], timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == b'' and b'no result after --rank_timeout_s 8' in r.stderr


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """the driver's own form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- ranks come from the launcher's environment (two of them sharing
    the one GPU of this box over gloo); the line must say so and every rank must be gated"""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    env['HND_DEFER_FPN'] = '0'          # two processes time-share one GPU here (bench.py's own launcher sets this itself)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--share_device', '--dist_backend', 'gloo', '--steps', '2', '--warmup', '1',
                        '--no_cpu_baseline'], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks']['launched_by'] == 'external launcher'
    assert out['loss_check']['ranks_gated'] == 2 and out['loss_check']['worst_rel_err_all_ranks'] < 1e-3
    assert out['ranks']['exchange_ms_per_step'] is not None
