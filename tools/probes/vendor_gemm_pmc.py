"""The library NT GEMM behind F.linear next to our k-tiled GEMM on the same operands, for a rocprofv3 --kernel-trace --pmc
pass (profile the vendor kernel that sustains 0.61-0.64 of peak under the same power cap).
    python tools/probes/vendor_gemm_pmc.py            # timings only
    rocprofv3 --kernel-trace --pmc ... -- python tools/probes/vendor_gemm_pmc.py --pmc   # few launches, long enough to reach the sustained clock
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd.compression.quantization import awq_ops

pmc = '--pmc' in sys.argv
shapes = [(65536, 14336, 4096), (65536, 4096, 14336), (2048, 4096, 4096)]
for (N, K, R) in shapes:
    gen = torch.Generator(device='cuda').manual_seed(N + K)
    x = (torch.randn(N, K, device='cuda', generator=gen) * torch.exp(0.5 * torch.randn(K, device='cuda', generator=gen))).to(torch.bfloat16)
    w = (torch.randn(R, K, device='cuda', generator=gen) * 0.02).to(torch.bfloat16)
    xt, wt = awq_ops.ktile_pack(x), awq_ops.ktile_pack(w)
    fl = 2.0 * N * K * R
    reps = 12 if N > 4096 else 200
    for name, fn in (('vendor F.linear', lambda: torch.nn.functional.linear(x, w)),
                     ('ours k-tiled', lambda: awq_ops.linear_out(xt, wt, None, tiled=True))):
        for _ in range(reps):          # warm: sustained clock
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f'N={N} K={K} R={R} {name:16s} {ms:8.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s = {fl / ms / 1e9 / 2500:.3f} of peak', flush=True)
    del x, w, xt, wt
