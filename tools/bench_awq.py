"""AWQ W4A16 g128 scale search + clip search on Llama-3-8B Linear shapes (BASELINE.json configs[2]) and FP8 e4m3
per-tensor quantization on Mixtral expert shapes (configs[4]); prints per-stage times and rates on MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llmc_amd.compression.quantization import FloatQuantizer, IntegerQuantizer, awq_ops
from llmc_amd.compression.quantization.awq_pipeline import search_scale_stacked


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    N, K = 128 * 512, 4096
    c = torch.exp(0.5 * torch.randn(K, generator=g, device=dev))
    c[torch.randperm(K, generator=g, device=dev)[:8]] *= 100
    x = (torch.randn(N, K, generator=g, device=dev) * c).to(torch.bfloat16)
    wq = IntegerQuantizer(4, True, 'per_group', group_size=128)
    for name, Rs in (('q|k|v', [4096, 1024, 1024]), ('o', [4096]), ('gate|up', [14336, 14336])):
        ws = [(torch.randn(R, K, generator=g, device=dev) * 0.02).to(torch.bfloat16) for R in Rs]
        R = sum(Rs)
        t = timed(lambda: search_scale_stacked(ws, x, wq, 'v2'))
        fl = 21 * 2.0 * N * R * K
        print(f'AWQ scale search {name:8s} N={N} K={K} R={R}: {t*1e3:8.2f} ms  {fl/t/1e12:7.1f} TFLOP/s over the 21 '
              f'evaluations ({fl/t/2.5e15*100:.1f}% of 2.5 PF)', flush=True)
        wcat = torch.cat(ws, 0)
        tg = timed(lambda: awq_ops.linear_out(x, wcat))
        print(f'    linear_eval alone: {tg*1e3:.3f} ms  {2.0*N*R*K/tg/1e12:.1f} TFLOP/s', flush=True)
    w = (torch.randn(4096, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    xs = x[0::N // 512].contiguous()
    t = timed(lambda: awq_ops.clip_search(w, xs, wq, True))
    macs = 11.0 * 4096 * K * xs.shape[0]
    print(f'AWQ clip search 4096x{K}, {xs.shape[0]} tokens: {t*1e3:.2f} ms  {macs/t/1e12:.2f} T rounded-MAC/s', flush=True)
    # FP8 per-tensor on Mixtral expert shapes
    fq = FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True)
    for (R, Kk) in ((14336, 4096), (4096, 14336)):
        w = (torch.randn(R, Kk, generator=g, device=dev) * 0.02).to(torch.bfloat16)
        t = timed(lambda: fq.real_quant_weight_dynamic(w))
        b = 5.0 * R * Kk
        print(f'FP8 e4m3 per-tensor {R}x{Kk}: {t*1e6:.1f} us  {b/t/1e12:.2f} TB/s algorithmic (5RK bytes)', flush=True)
    # RTN W4 g128 fake quant + real quant + pack
    w = (torch.randn(4096, 4096, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    from llmc_amd.compression.quantization import pack_lsb
    t = timed(lambda: wq.fake_quant_weight_dynamic(w))
    print(f'RTN W4 g128 fake-quant 4096x4096: {t*1e6:.1f} us  {4.0*4096*4096/t/1e12:.2f} TB/s', flush=True)
    def rq():
        codes, s, _ = wq.real_quant_weight_dynamic(w)
        return pack_lsb(codes, 4)
    t = timed(rq)
    print(f'RTN W4 g128 real-quant + pack 4096x4096: {t*1e6:.1f} us', flush=True)


if __name__ == '__main__':
    main()
