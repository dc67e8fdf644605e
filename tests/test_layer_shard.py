"""Multi-rank orchestration of the layer-sharded path, world_size 2 over Gloo on CPU. The compute callback is a
stand-in (the product's kernels need a GPU); what is checked is ownership, the broadcast and the ordered gather."""
import pytest
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from llmc_amd.dist import layer_shard as LS
    assert LS.world_info() == (rank, world)
    # independent units: every unit exactly once, in order, on rank 0
    res = LS.run_independent(7, lambda u: {'unit': u, 'rank': rank, 'w': torch.full((2, 2), float(u))})
    if rank == 0:
        assert [r['unit'] for r in res] == list(range(7))
        assert [r['rank'] for r in res] == [u % world for u in range(7)]
        assert all(float(r['w'][0, 0]) == r['unit'] for r in res)
    # cooperative subset: rank 1 owns the activations, everybody receives the same tensor
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4) if rank == 1 else None
    meta = ((2, 3, 4), torch.float32, 'cpu')
    out = LS.run_block_cooperative(['q', 'k', 'v'], x, 1, lambda name, xs: (name, float(xs.sum()), rank), meta)
    if rank == 0:
        assert [o[0] for o in out] == ['q', 'k', 'v'] and all(o[1] == 276.0 for o in out)
        assert [o[2] for o in out] == [0, 1, 0]
    # share the Hessian instead of the activations
    out = LS.run_block_cooperative(['gate', 'up'], x, 1, lambda name, h: (name, tuple(h.shape), float(h.trace())), meta,
                                   share='hessian', hessian_fn=lambda t: t.reshape(-1, 4).T @ t.reshape(-1, 4))
    if rank == 0:
        xr = torch.arange(24, dtype=torch.float32).reshape(-1, 4)
        assert all(o[1] == (4, 4) and abs(o[2] - float((xr.T @ xr).trace())) < 1e-3 for o in out)
    # tensors travel point-to-point (the RCCL path of bench.py --mode cooperative), nested payloads survive
    res = LS.run_independent(5, lambda u: {'w': torch.full((3,), float(u)), 'meta': [u, None, (torch.ones(2) * rank,)]},
                             to_cpu=False)
    if rank == 0:
        assert [float(r['w'][0]) for r in res] == [0.0, 1.0, 2.0, 3.0, 4.0]
        assert [r['meta'][0] for r in res] == list(range(5)) and all(r['meta'][1] is None for r in res)
        assert [float(r['meta'][2][0][0]) for r in res] == [float(u % world) for u in range(5)]
    # sample-sharded subset: each rank's Hessian of its own sequences, one all_reduce, row-sharded quantization
    K, R = 8, 40
    xs = torch.arange(4 * 6 * K, dtype=torch.float32).reshape(4, 6, K) / 100.0
    mine = xs[rank::world]
    ws = [torch.arange(R * K, dtype=torch.float32).reshape(R, K)[:24], torch.arange(R * K, dtype=torch.float32).reshape(R, K)[24:]]
    seen = {}

    def quant_rows(weights, H, rows):
        seen['H'] = H.clone()
        return torch.cat(weights, 0)[rows[0]:rows[1]] * 2
    out = LS.run_subset_sample_sharded(
        mine, ws, lambda x: (x.reshape(-1, K).T @ x.reshape(-1, K)) * (2.0 / x.shape[0]), quant_rows)
    full = (xs.reshape(-1, K).T @ xs.reshape(-1, K)) * (2.0 / 4)
    assert torch.allclose(seen['H'], full, rtol=1e-6)                    # mean of per-rank Hessians = global Hessian
    r0, r1 = out['rows']
    assert (r0, r1) == LS.row_range(rank, world, R) and torch.equal(out['payload'], torch.cat(ws, 0)[r0:r1] * 2)
    parts = [None] * world
    dist.all_gather_object(parts, (r0, r1))
    assert parts[0][0] == 0 and parts[-1][1] == R and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, 'ok'))


def test_layer_shard_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, 'ok'), (1, 'ok')]


def test_ownership_is_a_partition():
    from llmc_amd.dist.layer_shard import owner_of, units_of
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 224):
            seen = sorted(u for r in range(world) for u in units_of(r, n, world))
            assert seen == list(range(n))
            assert all(owner_of(u, world) == r for r in range(world) for u in units_of(r, n, world))
            sizes = [len(units_of(r, n, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('model', ['llama3-8b', 'llama3-70b'])
@pytest.mark.parametrize('mode', ['independent', 'handoff', 'cooperative'])
def test_per_rank_memory_plan_fits_an_mi355x(model, mode):
    """BASELINE configs[3] before there is an 8-GPU node to run it on: per-rank memory of a step from the library's own
    workspace queries (70B down_proj: 14 GiB of activations, a 3.1-GiB Hessian, its factor workspace, the partial tiles)."""
    import bench
    from llmc_amd.dist.layer_shard import plan_memory
    groups = bench.block_groups(model)
    plan = plan_memory(groups, 128, 2048, 8, mode)
    gib = {k: v / 2 ** 30 for k, v in plan.items()}
    assert plan['total'] < 0.85 * 288e9, gib
    if model == 'llama3-70b':
        assert gib['activations'] > (10 if mode != 'cooperative' else 4) and gib['hessians'] > 7, gib
    one = plan_memory(groups, 128, 2048, 1, 'independent')
    assert one['total'] >= plan['total'] - plan['handoff'] - 1 or mode != 'cooperative'
