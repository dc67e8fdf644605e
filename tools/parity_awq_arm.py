"""One ARM of the AWQ scale-search parity check at the shapes BASELINE configs[2] is quoted on:
`Awq.search_scale_subset` (awq.py:179-278; 20-point grid, one batch of N tokens, inspect = the subset's Linear layers)

  ref_rocm  the UNMODIFIED reference class (oracle/_ref_gpu) on this GPU through PyTorch-ROCm
  ref_cpu   the reference after its own CI rewrite (oracle/_ref) on the host cores (small shapes only)
  ours      llmc_amd.compression.quantization.awq_pipeline.search_scale_stacked

on the same seeded weights / activations (SURVEY 8d recipe, generated on the device from the seed: nothing large travels
between the arms). Output npz: the 20 losses, the argmin, best scales, x_mean, w_max. Test infrastructure only."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synth(Rs, K, N, seed, dev, dtype):
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    ws = []
    for R in Rs:
        w = torch.randn((R, K), generator=g, device=dev) * 0.02
        w[:, torch.randperm(K, generator=g, device=dev)[:max(1, K // 1000)]] *= 20.0
        ws.append(w.to(dtype))
    c = torch.exp(0.5 * torch.randn(K, generator=g, device=dev))
    c[torch.randperm(K, generator=g, device=dev)[:8]] *= 100.0
    x = torch.empty((N, K), device=dev, dtype=dtype)
    step = max(1, (1 << 26) // K)
    for i in range(0, N, step):
        n = min(step, N - i)
        x[i:i + n] = (torch.randn((n, K), generator=g, device=dev) * c).to(dtype)
    return ws, x


def run_reference(a, ws, x, dev):
    import torch
    import torch.distributed as dist
    from llmc.compression.quantization.awq import Awq
    from llmc.compression.quantization.quant import IntegerQuantizer
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29613')
        dist.init_process_group('nccl' if dev.type == 'cuda' else 'gloo', rank=0, world_size=1)

    class Stacked(torch.nn.Module):
        def __init__(self, layers):
            super().__init__()
            self.layers = torch.nn.ModuleList(layers)

        def forward(self, x):
            return torch.cat([l(x) for l in self.layers], dim=-1)

    if dev.type == 'cpu':      # awq.py:199 `v.cpu()` must copy like it does from a GPU (SURVEY 8c caveat iii)
        torch.Tensor.cpu = lambda self, *q, **k: self.clone()
    wq = IntegerQuantizer(4, a.sym, 'per_group', group_size=128)
    o = Awq.__new__(Awq)
    o.wquantizer, o.aquantizer, o.w_only, o.awq_bs, o.save_mem, o.padding_mask = wq, None, True, None, False, None
    o.trans_version, o.n_samples, o.has_gqa, o.do_gqa_trans = a.version, 1, False, False
    layers = []
    for w in ws:
        l = torch.nn.Linear(w.shape[1], w.shape[0], bias=False).to(w.dtype)
        l.weight.data = w
        layers.append(l.to(dev))
    losses = []
    orig = o.calculate_loss

    def rec(org_out, out, _orig=orig):
        v = _orig(org_out, out)
        losses.append(float(v))
        return v
    o.calculate_loss = rec
    layers_dict = {f'l{i}': l for i, l in enumerate(layers)}
    xb = x.unsqueeze(0)
    w_max = o.get_weight_scale(layers_dict)
    o._bs = 1
    x_mean = o.get_act_scale(xb)
    best = o.search_scale_subset(None, layers_dict, [xb], Stacked(layers), False, {})
    import numpy as np
    return {'losses': np.array(losses, dtype=np.float64), 'best': best.float().cpu().numpy(), 'w_max': w_max.float().cpu().numpy(),
            'x_mean': x_mean.float().cpu().numpy(), 'argmin': np.array(int(np.argmin(losses)))}


def run_ours(a, ws, x):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization import awq_ops
    from llmc_amd.compression.quantization.awq_pipeline import search_scale_stacked
    wq = IntegerQuantizer(4, a.sym, 'per_group', group_size=128)
    best, losses, n = search_scale_stacked(ws, x, wq, a.version, return_losses=True)
    w_max = None
    for w in ws:
        m = awq_ops.weight_mean(w, 128)
        w_max = m if w_max is None else w_max.add_(m)
    w_max = w_max.div_(len(ws))
    chunked = int(x.shape[0]) * sum(w.shape[0] for w in ws) * 2 > (1 << 32) - (1 << 20)
    return {'losses': losses.double().cpu().numpy(), 'best': best.float().cpu().numpy(), 'w_max': w_max.float().cpu().numpy(),
            'x_mean': awq_ops.act_mean(x).float().cpu().numpy(), 'argmin': np.array(int(n)), 'row_chunked': np.array(int(chunked))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arm', required=True, choices=['ref_rocm', 'ref_cpu', 'ours'])
    ap.add_argument('--rows', required=True, help='comma list: output rows of the stacked layers')
    ap.add_argument('--K', type=int, required=True)
    ap.add_argument('--N', type=int, required=True)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--sym', type=int, default=1)
    ap.add_argument('--version', default='v2')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    a.sym = bool(a.sym)
    if a.arm == 'ref_cpu':
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_shims'))
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    elif a.arm == 'ref_rocm':
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_shims'))
        sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref_gpu'))
    import numpy as np
    import torch
    gpu = torch.cuda.is_available()
    gen_dev = torch.device('cuda', 0) if gpu else torch.device('cpu')
    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    ws, x = synth([int(r) for r in a.rows.split(',')], a.K, a.N, a.seed, gen_dev, dtype)
    t0 = time.perf_counter()
    if a.arm == 'ours':
        out = run_ours(a, ws, x)
    else:
        dev = torch.device('cpu') if a.arm == 'ref_cpu' else gen_dev
        out = run_reference(a, [w.to(dev) for w in ws], x.to(dev), dev)
    if gpu:
        torch.cuda.synchronize()
    out['t_total'] = np.array(time.perf_counter() - t0)
    np.savez(a.out, **out)
    print(a.arm, a.rows, a.K, a.N, 'argmin', int(out['argmin']), 'done in %.1f s' % float(out['t_total']), flush=True)


if __name__ == '__main__':
    main()
