"""Layer-sharded quantization across the GPUs of one node: one process per GPU (torchrun), torch.distributed
over RCCL (backend 'nccl' on ROCm) / Gloo in CPU tests.

The hot path partitions naturally: given the calibration activations of a subset, every Linear
(and every output row) is independent. Two modes:

  * independent units (bench.py, synthetic layers, MoE experts, already-captured per-block activations): unit u is
    owned by rank u % world — no data-path collective at all; results are gathered to rank 0 for saving;
  * one block, several ranks: the rank that produced a subset's input broadcasts it ONCE (xGMI: a 2 GiB
    [128, 2048, 4096] bf16 tensor is ~14 ms on one 153 GB/s link, less as scatter + all-gather), each rank runs
    Hessian -> factor -> column loop for the layers it owns, results are gathered. When all layers of a subset
    share the input, it is cheaper to broadcast the 64 MiB Hessian instead (`share='hessian'`).

The reference's own multi-GPU mode (data-parallel calibration with a per-batch all_reduce of H, gptq.py:292) is
kept in GPTQ._sync_hessian with ONE reduction per Hessian.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def owner_of(unit, world):
    return unit % world


def units_of(rank, n_units, world):
    """Round-robin ownership: near-equal counts, consecutive units land on different GPUs."""
    return list(range(rank, n_units, world))


@dataclass
class ShardedResult:
    unit: int
    payload: object


def broadcast_tensor(t, src, shape=None, dtype=None, device=None):
    """Broadcast one tensor from `src` to every rank (receivers allocate from shape/dtype)."""
    rank, world = world_info()
    if world == 1:
        return t
    if rank != src:
        t = torch.empty(shape, dtype=dtype, device=device)
    dist.broadcast(t, src=src)
    return t


def run_independent(n_units, fn, gather_to=0):
    """Every rank runs fn(unit) for the units it owns; rank `gather_to` receives all results in unit order
    (other ranks get their own). `fn` returns any picklable / tensor payload (moved to CPU for the gather)."""
    rank, world = world_info()
    mine = [ShardedResult(u, fn(u)) for u in units_of(rank, n_units, world)]
    if world == 1:
        return [r.payload for r in mine]
    cpu = [ShardedResult(r.unit, _to_cpu(r.payload)) for r in mine]
    gathered = [None] * world if rank == gather_to else None
    dist.gather_object(cpu, gathered, dst=gather_to)
    if rank != gather_to:
        return [r.payload for r in mine]
    flat = sorted((r for part in gathered for r in part), key=lambda r: r.unit)
    assert [r.unit for r in flat] == list(range(n_units)), 'every unit exactly once'
    return [r.payload for r in flat]


def run_block_cooperative(layers, x_or_none, src, fn, x_meta, share='activations', hessian_fn=None, gather_to=0):
    """One subset, several ranks. `layers`: list of layer ids of the subset (same order on every rank);
    `x_or_none`: the subset's input on rank `src` (None elsewhere); `x_meta` = (shape, dtype, device) for receivers.
    share='activations': broadcast X, every rank computes its own Hessian for the layers it owns;
    share='hessian': rank `src` computes H once with hessian_fn(X) and broadcasts H (K x K fp32) instead of X.
    fn(layer_id, shared_tensor) -> payload. Returns the gathered payloads in `layers` order on rank gather_to."""
    rank, world = world_info()
    if share == 'hessian':
        shared = hessian_fn(x_or_none) if rank == src else None
        K = x_meta[0][-1]
        shared = broadcast_tensor(shared, src, (K, K), torch.float32, x_meta[2])
    else:
        shared = broadcast_tensor(x_or_none, src, *x_meta)
    out = run_independent(len(layers), lambda i: fn(layers[i], shared), gather_to=gather_to)
    return out


def _to_cpu(p):
    if torch.is_tensor(p):
        return p.detach().cpu()
    if isinstance(p, dict):
        return {k: _to_cpu(v) for k, v in p.items()}
    if isinstance(p, (list, tuple)):
        return type(p)(_to_cpu(v) for v in p)
    return p
