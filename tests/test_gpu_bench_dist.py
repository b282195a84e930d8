"""The data-parallel launch form of bench.py on ONE rank over RCCL (GPODE_BENCH_FORCE_DIST=1): the whole step -- BatchNorm
all-gathers, gradient all-reduce and Adam included -- must capture into a HIP graph and replay.  (This is the path the multi-GPU
scaling run takes; an eager step on the default stream ahead of the capture once made hipStreamEndCapture segfault.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('mode', ['whole', 'fwdbwd'])
def test_one_rank_rccl_step_captures_and_replays(mode):
    env = dict(os.environ, GPODE_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'cfg1', '--no-extra', '--no-cpu-baseline', '--steps', '5',
                        '--warmup', '2', '--dp-graph', mode], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['config']['hip_graph'] is True and line['config']['graph_scope'] == mode, line['config']
    assert line['dist_backend'] == 'nccl' and line['rccl_ranks'] == 1
    coll = line['collectives_per_step']
    assert coll['all_reduce']['count'] == 1 and coll['all_gather']['count'] == 10, coll
    assert line['value'] > 0
