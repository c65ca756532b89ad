"""bench.py's own multi-rank flow (spawn -> shard the games by index -> play -> the ONE example gather to rank 0 -> rank-reduced JSON line),
run at world size 2 on however many GPUs the box has (one: the two ranks share it), collectives over gloo -- so that the first N > 1 run of
the driver is not the first time this code path executes.  SURVEY.md 8(e); Coach.py:150-215 consumes the examples on one rank (dst = 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_multi_rank_flow_world2():
    env = dict(os.environ, AZG_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--games', '64', '--sims', '32', '--steps', '40', '--warmup', '2',
           '--preroll-plies', '40', '--no-cpu-baseline', '--no-secondary', '--no-sustained', '--roofline-rounds', '0']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]                 # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['rccl_world'] == 2 and r['scaling'] == 'weak'
    assert r['engine_errors'] == 0 and r['examples_dropped'] == 0
    per_rank = r['examples_per_rank']
    assert len(per_rank) == 2 and all(c > 0 for c in per_rank), per_rank          # both ranks were seen by the count all_gather
    assert r['examples_gathered'] == sum(per_rank)                                   # rank 0 received exactly what the ranks sent
    assert r['gather_bytes_received_rank0'] == sum(per_rank) * r['gather_row_bytes']
    assert r['plies_completed'] > 2 * 64 * 20 and r['value'] > 0                      # both ranks' plies are in the whole-job figure
    assert abs(r['value'] * r['ms_per_step'] * r['steps'] / 1e3 - r['plies_completed']) < 1.0
    assert r['config']['games_per_gpu'] == 64
    # every rank's own rate, wall time and HBM margin are in the line (a straggler or a nearly full GPU shows up in SCALE_rNN.json)
    assert len(r['value_per_rank']) == 2 and all(v > 0 for v in r['value_per_rank'])
    assert abs(sum(r['value_per_rank']) - r['value']) / r['value'] < 0.2 and max(r['ms_per_step_per_rank']) <= r['ms_per_step'] * 1.001
    assert all(0 < f <= t for f, t in zip(r['free_hbm_bytes_per_rank'], r['total_hbm_bytes_per_rank']))


def test_bench_multi_rank_flow_fails_loudly():
    """a rank that dies takes the run down with a non-zero exit code and a message, not a silent hang or an empty success"""
    env = dict(os.environ, AZG_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1', AZG_BENCH_FAIL_RANK='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--games', '64', '--sims', '32', '--steps', '2', '--warmup', '1',
           '--preroll-plies', '0', '--no-cpu-baseline', '--no-secondary', '--no-sustained', '--roofline-rounds', '0']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode != 0
    assert 'no result line' in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
