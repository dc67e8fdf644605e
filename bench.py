#!/usr/bin/env python
"""bench.py — layers/sec of the GPTQ W4A16 hot path on Llama-3-8B Linear shapes (BASELINE.json metric).

One "step" = GPTQ of one transformer block's seven Linear layers (q,k,v,o,gate,up,down; four distinct
inputs) from 128 x 2048 resident calibration tokens: Hessian (MFMA) -> actorder/dead/damp -> Cholesky +
inverse -> blocked column loop -> scales/zeros/compensated weights.  Synthetic inputs.
N > 1: one process per GPU (torchrun), every rank quantizes its own blocks (layer-sharded, no data-path
collective); value = layers of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value + roofline + cpu_baseline.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15  # dense bf16/f16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md

MODELS = {
    # name: (hidden, kv_out, ffn, n_blocks)
    'llama3-8b': (4096, 1024, 14336, 32),
    'llama3-70b': (8192, 1024, 28672, 80),
    'tiny': (512, 128, 1024, 2),
}


def block_groups(model):
    h, kv, ffn, _ = MODELS[model]
    # (input name, K, [(layer name, R)]) — llmc/models/llama.py:52-91 subsets
    return [('attn_in', h, [('q_proj', h), ('k_proj', kv), ('v_proj', kv)]),
            ('o_in', h, [('o_proj', h)]),
            ('mlp_in', h, [('gate_proj', ffn), ('up_proj', ffn)]),
            ('down_in', ffn, [('down_proj', h)])]


def synth_weight(R, K, seed, device, dtype):
    g = torch.Generator(device=device).manual_seed(1000 + seed)
    w = torch.randn(R, K, generator=g, device=device, dtype=torch.float32) * 0.02
    n_out = max(1, K // 1000)
    idx = torch.randperm(K, generator=g, device=device)[:n_out]
    w[:, idx] *= 20.0
    return w.to(dtype)


def synth_acts(n_seq, seq, K, seed, device, dtype):
    g = torch.Generator(device=device).manual_seed(2000 + seed)
    c = torch.exp(0.5 * torch.randn(K, generator=g, device=device))
    idx = torch.randperm(K, generator=g, device=device)[:8]
    c[idx] *= 100.0
    x = torch.empty((n_seq, seq, K), device=device, dtype=dtype)
    step = max(1, min(n_seq, (1 << 28) // (seq * K)))
    for i in range(0, n_seq, step):
        z = torch.randn((min(step, n_seq - i), seq, K), generator=g, device=device, dtype=torch.float32)
        x[i:i + step] = (z * c).to(dtype)
    return x


def cpu_baseline(model, n_seq, seq, cfg):
    """The oracle (a CPU port of the reference path) timed on the host cores on a bounded sample."""
    import numpy as np

    from oracle import gptq_ref as G
    from oracle import quant_ref as Q
    cores = os.cpu_count() or 1
    h, kv, ffn, _ = MODELS[model]
    K = min(h, 4096)
    rng = np.random.RandomState(0)
    # Hessian: 4 sequences of the K-channel input (fp32 sgemm like the reference), scaled to n_seq
    nb = 4
    H = np.zeros((K, K), dtype=np.float32)
    n = 0
    t0 = time.time()
    for _ in range(nb):
        x = rng.standard_normal((seq, K)).astype(np.float32)
        H, n = G.add_batch(H, n, x)
    t_h_seq = (time.time() - t0) / nb
    H += np.eye(K, dtype=np.float32) * 0.1
    W = (rng.standard_normal((K, K)) * 0.02).astype(np.float32)
    t0 = time.time()
    perm = G.hessian_sorting(H)
    Wp, U = G.process_hessian_and_weights(W, H, perm, cfg.percdamp)
    t_chol = time.time() - t0
    qmin, qmax = Q.int_range(cfg.bit, cfg.symmetric)
    t0 = time.time()
    G.weight_transform(Wp, U, cfg.symmetric, qmin, qmax, cfg.group_size)
    t_loop = time.time() - t0
    # model of one block, as the reference executes it (7 Hessians / factorisations / loops per block),
    # scaling the measured K-wide pieces by their flop counts
    def hess(k):
        return t_h_seq * n_seq * (k / K) ** 2

    def chol(k):
        return t_chol * (k / K) ** 3

    def loop(r, k):
        return t_loop * (r / K) * (k / K) ** 2

    t_block = 0.0
    layers = 0
    for _, k, ls in block_groups(model):
        for _, r in ls:
            t_block += hess(k) + chol(k) + loop(r, k)
            layers += 1
    return {
        'value': layers / t_block, 'unit': 'layers/s', 'cores': cores, 'kind': 'port',
        'sample': (f'numpy/C oracle on {cores} host threads: Hessian = {nb} of {n_seq} sequences of one {K}-channel '
                   f'input ({t_h_seq:.3f} s/seq), factorisation ({t_chol:.2f} s) and column loop ({t_loop:.2f} s) of '
                   f'one {K}x{K} layer in full; other shapes scaled by flop count; 7 Hessians per block as the '
                   'reference executes them'),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--model', default='llama3-8b', choices=list(MODELS))
    ap.add_argument('--n-seq', type=int, default=128)
    ap.add_argument('--seq-len', type=int, default=2048)
    ap.add_argument('--calib-bs', type=int, default=128,
                    help='sequences per Hessian launch (reference calib.bs; 128 = one launch per input)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--variant', default='w_only', choices=['w_only', 'vllm'],
                    help='w_only: asym g128 actorder dynamic groups (configs/quantization/methods/GPTQ/gptq_w_only.yml); '
                         'vllm: sym g128 static groups + INT4 pack (configs/quantization/backend/vllm/gptq_w4a16.yml)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in llmc_amd)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    from llmc_amd.compression.quantization import IntegerQuantizer, pack_lsb
    from llmc_amd.compression.quantization.gptq_pipeline import GptqConfig, quantize_stacked
    from llmc_amd.compression.quantization.hessian import HessianAccumulator

    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    if args.variant == 'w_only':
        cfg = GptqConfig(bit=4, symmetric=False, group_size=128, actorder=True, static_groups=False)
    else:
        cfg = GptqConfig(bit=4, symmetric=True, group_size=128, actorder=True, static_groups=True)
    groups = block_groups(args.model)

    # ---- resident synthetic data (different seeds per rank: every rank owns different blocks)
    acts, weights, accs, hwork = {}, {}, {}, {}
    for gi, (name, K, layers) in enumerate(groups):
        acts[name] = synth_acts(args.n_seq, args.seq_len, K, rank * 64 + gi, dev, dtype)
        weights[name] = [synth_weight(R, K, rank * 64 + gi * 8 + li, dev, dtype) for li, (_, R) in enumerate(layers)]
        accs[name] = HessianAccumulator(K, dev)
        hwork[name] = torch.empty((K, K), dtype=torch.float32, device=dev)
    wq = IntegerQuantizer(cfg.bit, cfg.symmetric, 'per_group', group_size=cfg.group_size)
    n_layers_block = sum(len(ls) for _, _, ls in groups)

    timing = []

    def step(record):
        outs = []
        for name, K, layers in groups:
            acc = accs[name]
            acc.timing = timing if record else None
            acc.reset()
            x = acts[name]
            for i in range(0, args.n_seq, args.calib_bs):
                acc.add(x[i:i + args.calib_bs])
            static = None
            if cfg.static_groups:
                # collect_block_qparams (base_blockwise_quantization.py:338-365): RTN qparams of the original weights
                static = []
                for w in weights[name]:
                    _, s, z, _, _ = wq.get_tensor_qparams(w)
                    static.append((s, None if cfg.symmetric else z))
            res = quantize_stacked(weights[name], acc.H, cfg, static_qparams=static, h_work=hwork[name])
            if args.variant == 'vllm':
                for r, (s, _) in zip(res, static):
                    a = {'scales': s, 'zeros': torch.tensor(0.0), 'qmax': wq.qmax, 'qmin': wq.qmin}
                    codes, _, _ = wq.real_quant_weight_static(r.weight, a)     # GPTQ.w_q (gptq.py:412-422)
                    outs.append(pack_lsb(codes, cfg.bit))
            outs.append(res)
        return outs

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel (k_syrk), HIP events on the launch stream, this rank
    fl = sum(T * K * (K + 1) for (_, _, T, K) in timing)
    ms = sum(e0.elapsed_time(e1) for (e0, e1, _, _) in timing)
    n_launch = len(timing)
    achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    # HBM-side bytes of the dominant kernel come from PMC passes that cannot run inside the timed process
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/pmc_bench.sh); the committed summary is reported with its source.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    if args.model == 'llama3-8b' and args.n_seq == 128 and args.seq_len == 2048 and args.calib_bs == 128 \
            and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))['k_syrk']
            traffic, traffic_src = tj['hbm_bytes_per_launch'], 'profiles/r01_pmc_traffic.json (' + tj['note'] + ')'
        except Exception:
            traffic = None

    if rank == 0:
        total_layers = n_layers_block * args.steps * world
        out = {
            'metric': 'layers/sec (GPTQ W4A16, %s Linear shapes, %dx%d calib)' % (
                {'llama3-8b': 'Llama-3-8B', 'llama3-70b': 'Llama-3-70B'}.get(args.model, args.model), args.n_seq,
                args.seq_len),
            'value': total_layers / dt, 'unit': 'layers/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {
                'workload': (f'GPTQ W4A16 g128 ({args.variant}) on {args.model}-shaped random-init Linear layers, '
                             f'1 transformer block (7 Linear, 4 distinct inputs) per step per GPU'),
                'n_seq': args.n_seq, 'seq_len': args.seq_len, 'calib_bs': args.calib_bs,
                'symmetric': cfg.symmetric, 'actorder': cfg.actorder, 'static_groups': cfg.static_groups,
                'parallelism': f'layer-sharded x{world}' if world > 1 else 'single GPU',
            },
            'roofline': {
                'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_MFMA_16BIT / 1e12, 'unit': 'TFLOP/s',
                'frac': achieved * 1e12 / PEAK_MFMA_16BIT, 'traffic': traffic, 'traffic_source': traffic_src,
                'kernel': 'k_syrk (llmc_hessian_accum_partials)', 'launches': n_launch,
                'algorithmic_flops_per_launch': fl / max(1, n_launch),
                'avg_launch_ms': ms / max(1, n_launch),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(args.model, args.n_seq, args.seq_len, cfg)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out['cpu_baseline'] = {'value': None, 'unit': 'layers/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {type(e).__name__}: {e}'}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
