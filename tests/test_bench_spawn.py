"""bench.py's multi-rank plumbing without a GPU: `--gpus N --dry` self-spawns N ranks (Gloo; N = 2, 4, 8), runs the hand-off
(timed by default), independent and cooperative steps with CPU stand-ins, and prints the contract's JSON line with n_gpus = N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra, env_extra=None, gpus=2):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--dry', '--steps', '2',
                        '--warmup', '1'] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def check_contract(j, gpus):
    assert j['n_gpus'] == gpus and j['steps'] == 2 and j['warmup'] == 1
    assert j['unit'] == 'layers/s' and j['value'] > 0 and j['higher_is_better'] is True
    assert 'dry-run' in j['data']
    for k in ('metric', 'ms_per_step', 'vs_baseline', 'dtype', 'config', 'roofline'):
        assert k in j
    assert j['value_mode'] in ('independent_no_comm', 'handoff_gloo_dry', 'cooperative') and 'interrank_bytes_per_step_per_rank' in j


@pytest.mark.parametrize('mode', ['independent', 'cooperative', 'handoff'])
def test_self_spawn_two_ranks_dry(mode):
    j = run(['--mode', mode])
    check_contract(j, 2)
    assert j['scaling'] == ('strong' if mode == 'cooperative' else 'weak')
    if mode == 'cooperative':
        assert 'cooperative x2' in j['config']['parallelism']
    if mode == 'handoff':
        # block-sharded with the owner-to-owner hand-off of a block's output over send/recv
        assert 'handed owner-to-owner' in j['config']['parallelism']
    if mode == 'independent':
        assert 'no data-path traffic' in j['config']['parallelism']
    assert 'independent_value' not in j and 'cooperative_value' not in j        # an explicit mode measures that mode only
    assert j['value_mode'] == {'independent': 'independent_no_comm', 'handoff': 'handoff_gloo_dry', 'cooperative': 'cooperative'}[mode]


@pytest.mark.parametrize('gpus', [2, 4, 8])
def test_default_mode_times_the_handoff_partition_and_carries_the_others(gpus):
    """No --mode with N > 1 (what the driver's scaling run launches): one hand-off step is tried first on every rank; it works
    here, so the TIMED region runs north_star's partition (every rank owns its blocks, a block's calibration activations arrive
    from the previous owner over send/recv: `value_mode: handoff...`), and ONE line also carries independent_value (the same
    ownership without traffic) and cooperative_value (one block shared by all ranks) measured outside the timed region. World
    sizes 4 and 8 run the same collectives the 8-GPU node will (Gloo here): broadcast, all_reduce, batched send/recv, ring."""
    j = run([], gpus=gpus)
    check_contract(j, gpus)
    assert j['value_mode'].startswith('handoff') and j['scaling'] == 'weak'
    assert 'handed owner-to-owner' in j['config']['parallelism']
    assert j['independent_value'] > 0 and j['cooperative_value'] > 0
    assert 'handoff_error' not in j and 'cooperative_error' not in j
    assert j['interrank_bytes_per_step_per_rank']['handoff'] > 0 and j['interrank_bytes_per_step_per_rank']['independent'] == 0


@pytest.mark.parametrize('gpus', [4, 8])
def test_cooperative_mode_at_node_world_sizes(gpus):
    j = run(['--mode', 'cooperative'], gpus=gpus)
    check_contract(j, gpus)
    assert j['scaling'] == 'strong' and f'cooperative x{gpus}' in j['config']['parallelism']


def test_failing_handoff_is_recorded_and_does_not_take_the_line_down():
    """The safety net of the first multi-GPU run. Default mode: the pre-flight hand-off step raising (on every rank, like an
    unavailable peer-to-peer path) makes all ranks agree — over the Gloo control group — to time the same ownership WITHOUT
    data-path traffic: the contract value and cooperative_value are still there, the reason is in handoff_error. Explicit --mode
    handoff: the same agreement."""
    j = run([], {'LLMC_BENCH_DRY_FAIL_HANDOFF': '1'})
    assert j['n_gpus'] == 2 and j['value'] > 0 and j['scaling'] == 'weak'
    assert j['value_mode'] == 'independent_no_comm' and 'no data-path traffic' in j['config']['parallelism']
    assert 'handoff_error' in j and 'independent_value' not in j and j['cooperative_value'] > 0
    j = run(['--mode', 'handoff'], {'LLMC_BENCH_DRY_FAIL_HANDOFF': '1'})
    assert j['value'] > 0 and 'handoff_error' in j and 'no data-path traffic' in j['config']['parallelism']
    assert j['value_mode'] == 'independent_no_comm'


def test_single_rank_needs_a_gpu_or_dry():
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1'], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'needs an MI355X' in (r.stdout + r.stderr)
