"""`python bench.py --gpus N` must itself start N fresh rank processes (the driver's form of the command has no
torchrun in front of it), relay rank 0's JSON line and report n_gpus == N.  Exercised here on the CPU with the
launcher's dry-run step (GPODE_BENCH_DRYRUN=1: rendezvous, barrier, max-over-ranks timing over gloo; no kernels), and under
`torch.distributed.run` -- the driver's other form -- with the same flag."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    e.update(GPODE_BENCH_DRYRUN='1', **kw)
    return e


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.timeout(180)
@pytest.mark.parametrize('n', [1, 2, 3])
def test_gpus_flag_starts_n_ranks(n):
    r = subprocess.run([sys.executable, BENCH, '--gpus', str(n), '--steps', '3', '--warmup', '1'], env=_env(), capture_output=True,
                       text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == n and out['dist_ranks'] == n and out['steps'] == 3 and out['dry_run'] is True
    assert out['config']['parallelism'] == 'dp%d' % n


@pytest.mark.timeout(180)
def test_under_torchrun_the_flag_must_match_world_size():
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
            '--master-port', '0', BENCH, '--steps', '2', '--warmup', '0']
    port = subprocess.run([sys.executable, '-c', 'import socket;s=socket.socket();s.bind(("127.0.0.1",0));print(s.getsockname()[1])'],
                          capture_output=True, text=True).stdout.strip()
    base[base.index('0', base.index('--master-port'))] = port
    ok = subprocess.run(base + ['--gpus', '2'], env=_env(), capture_output=True, text=True, timeout=150)
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert _one_json_line(ok.stdout)['n_gpus'] == 2
    bad = subprocess.run(base + ['--gpus', '4'], env=_env(), capture_output=True, text=True, timeout=150)
    assert bad.returncode != 0 and 'WORLD_SIZE=2' in bad.stderr


@pytest.mark.timeout(120)
def test_a_failing_rank_fails_the_job():
    # rank 1 of 2 is told to exit early: the launcher must stop rank 0 (blocked in the rendezvous) and return non-zero
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '0'], env=_env(GPODE_BENCH_DRYRUN_FAIL_RANK='1'),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode != 0
    assert 'rank 1 exited with code' in r.stderr
