"""The N > 1 leg of bench.py on CPU: two gloo ranks, barrier + max-over-ranks timing, aggregate value.
(The PME path itself is "replicas only" across GPUs this round -- DESIGN.md -- so the only collective
is the timing reduction.)"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    dist.barrier()
    dt = 0.5 + 0.25 * rank                      # rank 1 is the slow one
    t = bench.reduce_max_seconds(dt, dist, 'cpu')
    assert abs(t - 0.75) < 1e-12, t
    v = bench.aggregate_ns_per_day(t / 10, world)
    assert abs(v - 2 * 0.0864 / 0.075) < 1e-9, v
    dist.barrier()
    if rank == 0:
        print('OK', world, v)
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_timing_reduction(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'OK 2' in r.stdout
