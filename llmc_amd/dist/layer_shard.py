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


def run_independent(n_units, fn, gather_to=0, to_cpu=True):
    """Every rank runs fn(unit) for the units it owns; rank `gather_to` receives all results in unit order
    (other ranks get their own). `fn` returns a payload: tensors / None / numbers nested in dicts, lists, tuples.
    to_cpu=True: payloads travel as pickled CPU objects (gather_object; any backend, what the Gloo tests use).
    to_cpu=False: tensors stay on their device and travel point-to-point (RCCL send/recv over xGMI); only their
    shapes / dtypes are exchanged as objects. gather_to=None: no gather at all, every rank keeps what it computed."""
    rank, world = world_info()
    mine = [ShardedResult(u, fn(u)) for u in units_of(rank, n_units, world)]
    if world == 1 or gather_to is None:      # gather_to=None: results stay with their owners (saved / gathered later)
        return [r.payload for r in mine]
    if not to_cpu:
        return _gather_p2p(mine, n_units, gather_to)
    cpu = [ShardedResult(r.unit, _to_cpu(r.payload)) for r in mine]
    gathered = [None] * world if rank == gather_to else None
    dist.gather_object(cpu, gathered, dst=gather_to)
    if rank != gather_to:
        return [r.payload for r in mine]
    flat = sorted((r for part in gathered for r in part), key=lambda r: r.unit)
    assert [r.unit for r in flat] == list(range(n_units)), 'every unit exactly once'
    return [r.payload for r in flat]


def _flatten(p, out):
    """payload -> skeleton with tensor placeholders ('T', index); tensors appended to `out`."""
    if torch.is_tensor(p):
        out.append(p)
        return ('T', len(out) - 1)
    if isinstance(p, dict):
        return {k: _flatten(v, out) for k, v in p.items()}
    if isinstance(p, (list, tuple)):
        return type(p)(_flatten(v, out) for v in p)
    return p


def _unflatten(sk, tensors):
    if isinstance(sk, tuple) and len(sk) == 2 and sk[0] == 'T' and isinstance(sk[1], int):
        return tensors[sk[1]]
    if isinstance(sk, dict):
        return {k: _unflatten(v, tensors) for k, v in sk.items()}
    if isinstance(sk, (list, tuple)):
        return type(sk)(_unflatten(v, tensors) for v in sk)
    return sk


def _gather_p2p(mine, n_units, gather_to):
    rank, world = world_info()
    backend = dist.get_backend()
    flat = []
    for r in mine:
        ts = []
        sk = _flatten(r.payload, ts)
        if backend == 'nccl':
            for t in ts:
                assert t.is_cuda, 'layer_shard: RCCL sends device tensors only (a payload tensor lives on the CPU)'
        flat.append((r.unit, sk, [t.contiguous() for t in ts]))
    meta = [(u, sk, [(tuple(t.shape), t.dtype) for t in ts]) for u, sk, ts in flat]
    metas = [None] * world
    dist.all_gather_object(metas, meta)
    if rank != gather_to:
        ops = [dist.P2POp(dist.isend, t, gather_to) for u, sk, ts in flat for t in ts]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return [r.payload for r in mine]
    # receive buffers live where the backend moves data: the current GPU under RCCL (also when this rank owns no
    # unit and has no payload tensor of its own to look at), the host under Gloo
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    results = {u: p.payload for u, p in ((r.unit, r) for r in mine)}
    ops, pending = [], []
    for src in range(world):
        if src == gather_to:
            continue
        for u, sk, shapes in metas[src]:
            ts = [torch.empty(shp, dtype=dt, device=dev) for shp, dt in shapes]
            ops.extend(dist.P2POp(dist.irecv, t, src) for t in ts)
            pending.append((u, sk, ts))
    if ops:                                   # all transfers of all ranks in flight at once (one group call under RCCL)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for u, sk, ts in pending:
        results[u] = _unflatten(sk, ts)
    assert sorted(results) == list(range(n_units)), 'every unit exactly once'
    return [results[u] for u in range(n_units)]


def run_block_cooperative(layers, x_or_none, src, fn, x_meta, share='activations', hessian_fn=None, gather_to=0,
                          to_cpu=True):
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
    out = run_independent(len(layers), lambda i: fn(layers[i], shared), gather_to=gather_to, to_cpu=to_cpu)
    return out


def row_range(rank, world, n_rows, align=16):
    """Rows [r0, r1) of a stacked weight matrix owned by `rank`: near-equal, boundaries multiples of `align`."""
    per = -(-n_rows // world)
    per = -(-per // align) * align
    r0 = min(n_rows, rank * per)
    return r0, min(n_rows, r0 + per)


def run_subset_sample_sharded(x_local, weights, hessian_fn, quantize_rows_fn):
    """One subset, several ranks, calibration samples sharded (the reference's data-parallel semantics,
    gptq.py:292-295, and SURVEY.md §8e's plan for the widest inputs): every rank accumulates the Hessian of ITS
    sequences (each rank holds the same number), ONE all_reduce replaces the reference's per-batch one, every rank
    factors the identical H (same kernels, same bits), and the rows of the stacked weights — independent given
    Hinv — are split across ranks. Returns this rank's rows' payload and the row range; the caller gathers or saves.
    hessian_fn(x_local) -> H [K, K] fp32 (scaled by 2 / local sequences); quantize_rows_fn(weights, H, (r0, r1))."""
    rank, world = world_info()
    H = hessian_fn(x_local)
    if world > 1:
        dist.all_reduce(H, op=dist.ReduceOp.SUM)
        H.div_(world)
    n_rows = sum(int(w.shape[0]) for w in weights)
    rows = row_range(rank, world, n_rows)
    payload = quantize_rows_fn(weights, H, rows) if rows[1] > rows[0] else None
    return {'rows': rows, 'payload': payload}


def _to_cpu(p):
    if torch.is_tensor(p):
        return p.detach().cpu()
    if isinstance(p, dict):
        return {k: _to_cpu(v) for k, v in p.items()}
    if isinstance(p, (list, tuple)):
        return type(p)(_to_cpu(v) for v in p)
    return p


def plan_memory(groups, n_seq, seq_len, world, mode, act_bytes=2):
    """Per-rank device memory (bytes, the worst rank) of one bench / quantization step, by item. `groups` = [(input name, K,
    [(layer, R), ...]), ...] as bench.block_groups; mode: 'independent' | 'handoff' | 'cooperative'. Workspace sizes come from
    the library's own `*_ws_bytes` queries (pure host calls), so this runs without a GPU: the check that BASELINE configs[3]
    (Llama-3-70B shapes, 8 x MI355X) fits 288 GB per GPU before anyone has an 8-GPU node to try it on.
      activations   the resident calibration input of every distinct subset input (sample-sharded subsets hold n_seq / world)
      hessians      H [K, K] fp32 per subset + the permuted / damped work copy
      syrk partial  k_syrk4's partial tiles of the widest launch
      factor ws     llmc_chol_inv_upper workspace of the widest subset (one per concurrently factored subset)
      loop ws       llmc_gptq_quantize workspace + fp32 stacked weights in / out + losses
      hand-off      the three [n_seq, seq, hidden] buffers of --mode handoff"""
    from llmc_amd import _ffi
    L = _ffi.lib()
    T = n_seq * seq_len
    items = {'activations': 0, 'hessians': 0, 'syrk_partials': 0, 'factor_ws': 0, 'loop_ws': 0, 'weights': 0, 'handoff': 0}
    coop = mode == 'cooperative' and world > 1
    for name, K, layers in groups:
        R = sum(r for _, r in layers)
        sample = coop and K > 8192
        n_mine = -(-n_seq // world) if sample else n_seq
        # every rank holds its inputs: its own sequences (sample-sharded), or the whole set (rank 0 holds the broadcast inputs and
        # the receivers allocate the same size)
        items['activations'] += n_mine * seq_len * K * act_bytes
        items['hessians'] += 2 * K * K * 4
        items['syrk_partials'] = max(items['syrk_partials'], int(L.llmc_hessian_accum_ws_bytes(n_mine * seq_len, K, K)))
        items['factor_ws'] += int(L.llmc_chol_inv_upper_ws_bytes(K)) + K * K * 4       # the four subsets' chains run side by side
        rows = -(-R // world) if sample else R
        items['loop_ws'] += int(L.llmc_gptq_quantize_ws_bytes(rows, K)) + 3 * rows * K * 4
        items['weights'] += R * K * act_bytes
    if mode == 'handoff' and world > 1:
        items['handoff'] = 3 * T * groups[0][1] * act_bytes
    items['total'] = sum(items.values())
    return items
