"""Worker of tests/test_rccl_gpu.py: one rank per GPU over RCCL. Checks the cooperative modes of
llmc_amd/dist/layer_shard.py against the single-GPU result computed by rank 0 on the same data."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    from llmc_amd.compression.quantization.gptq_pipeline import GptqConfig, quantize_stacked
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    from llmc_amd.dist import layer_shard as LS
    cfg = GptqConfig(bit=4, symmetric=False, group_size=128, actorder=True, static_groups=False)
    K, n_seq, seq = 1024, 8, 512
    rows = [512, 128, 256]
    g = torch.Generator(device=dev).manual_seed(7)                      # same seed: same data on every rank
    X = (torch.randn(n_seq, seq, K, generator=g, device=dev) * torch.exp(0.5 * torch.randn(K, generator=g, device=dev))).to(torch.bfloat16)
    Ws = [(torch.randn(r, K, generator=g, device=dev) * 0.02).to(torch.bfloat16) for r in rows]

    def hessian(x):
        acc = HessianAccumulator(K, dev)
        for i in range(x.shape[0]):
            acc.add(x[i:i + 1])
        return acc.H.clone()

    def quant(li, shared, is_h):
        H = shared if is_h else hessian(shared)
        r = quantize_stacked([Ws[li]], H.clone(), cfg)[0]
        return {'weight': r.weight, 'scales': r.scales, 'zeros': r.zeros}

    meta = ((n_seq, seq, K), torch.bfloat16, dev)
    single = [quant(li, X, False) for li in range(len(rows))] if rank == 0 else None
    for share in ('activations', 'hessian'):
        out = LS.run_block_cooperative(list(range(len(rows))), X if rank == 0 else None, 0,
                                       lambda li, sh, s=share: quant(li, sh, s == 'hessian'), meta, share=share,
                                       hessian_fn=hessian, gather_to=0, to_cpu=False)
        if rank == 0:
            for a, b in zip(out, single):
                for k in ('weight', 'scales', 'zeros'):
                    assert torch.equal(a[k].to(dev), b[k]), (share, k)
    # sample-sharded: every rank's own sequences, ONE all_reduce, row-sharded column loop
    res = LS.run_subset_sample_sharded(X[rank::world].contiguous(), Ws, hessian,
                                       lambda ws, H, rr: quantize_stacked(ws, H.clone(), cfg, rows=rr)[0].weight)
    r0, r1 = res['rows']
    if rank == 0:
        full = torch.cat([s['weight'] for s in single], 0)
        d = (res['payload'] - full[r0:r1]).abs()
        # the mean of per-rank running means rounds differently from one running mean: statistical agreement
        assert float((d <= 1e-3 * full.abs().max()).float().mean()) > 0.9
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print('RCCL_OK', flush=True)


if __name__ == '__main__':
    main()
