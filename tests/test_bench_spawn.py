"""bench.py's multi-rank plumbing without a GPU: `--gpus 2 --dry` self-spawns two ranks (gloo), runs the independent, hand-off and
cooperative steps with CPU stand-ins, and prints the contract's JSON line with n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra, env_extra=None):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry', '--steps', '2',
                        '--warmup', '1'] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('mode', ['independent', 'cooperative', 'handoff', None])
def test_self_spawn_two_ranks_dry(mode):
    j = run(['--mode', mode] if mode else [])            # no flag: the N > 1 default = handoff
    assert j['n_gpus'] == 2 and j['steps'] == 2 and j['warmup'] == 1
    assert j['unit'] == 'layers/s' and j['value'] > 0 and j['higher_is_better'] is True
    assert j['scaling'] == ('strong' if mode == 'cooperative' else 'weak')
    assert 'dry-run' in j['data']
    for k in ('metric', 'ms_per_step', 'vs_baseline', 'dtype', 'config', 'roofline'):
        assert k in j
    if mode == 'cooperative':
        assert 'cooperative x2' in j['config']['parallelism']
    if mode in ('handoff', None):
        # block-sharded with the owner-to-owner hand-off of the calibration activations over send/recv, plus the same
        # ownership without the hand-off measured after the timed region
        assert 'handed owner-to-owner' in j['config']['parallelism'] and j['independent_value'] > 0
    if mode == 'independent':
        assert 'independent_value' not in j


def test_handoff_failure_falls_back_to_the_same_ownership_without_it():
    """The safety net of the first multi-GPU run: the transfer raising (on every rank, like an unavailable peer-to-peer path)
    makes the ranks agree to continue with the same ownership without the hand-off; the line carries the reason."""
    j = run([], {'LLMC_BENCH_DRY_FAIL_HANDOFF': '1'})
    assert j['n_gpus'] == 2 and j['value'] > 0 and j['scaling'] == 'weak'
    assert 'handoff_error' in j and 'no data-path traffic' in j['config']['parallelism']
    assert 'independent_value' not in j


def test_single_rank_needs_a_gpu_or_dry():
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1'], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'needs an MI355X' in (r.stdout + r.stderr)
