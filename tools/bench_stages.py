"""Per-stage wall time (HIP events) of the GPTQ pipeline for each Llama-3-8B subset shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llmc_amd import _ffi
if os.environ.get('LLMC_PROBE_LIB'):          # lab builds (tools/probes): A/B of one kernel file
    _ffi.LIB_PATH = os.environ['LLMC_PROBE_LIB']
from llmc_amd.compression.quantization import gptq_ops
from llmc_amd.compression.quantization.hessian import HessianAccumulator


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def run(name, K, Rs, T=65536):
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(T, K, generator=g, device=dev).to(torch.bfloat16)
    W = torch.cat([(torch.randn(R, K, generator=g, device=dev) * 0.02).to(torch.bfloat16) for R in Rs], 0)
    acc = HessianAccumulator(K, dev)
    for it in range(2):
        acc.reset()
        e0 = ev(); acc.add(x.unsqueeze(0))
        e1 = ev(); perm = torch.argsort(torch.diagonal(acc.H), descending=True)
        Hp, Wp = gptq_ops.hessian_prep(acc.H, W, perm, 0.01)
        e2 = ev(); U = gptq_ops.chol_inv_upper(Hp, check=False)
        e3 = ev(); tmp, losses, s, z = gptq_ops.gptq_quantize(Wp, U, False, 0.0, 15.0, 128)
        e4 = ev(); out = tmp.index_select(1, torch.argsort(perm)); ls = losses.sum()
        e5 = ev()
        torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ((e0, e1), (e1, e2), (e2, e3), (e3, e4), (e4, e5))]
    print(f'{name:8s} K={K} R={sum(Rs)}: hessian(T={T}) {t[0]:.2f} ms | prep {t[1]:.2f} | chol+inv {t[2]:.2f} | '
          f'column loop {t[3]:.2f} | unpermute+loss {t[4]:.2f}', flush=True)


if __name__ == '__main__':
    if '--70b' in sys.argv:                       # Llama-3-70B subset shapes (BASELINE configs[3])
        run('q|k|v', 8192, [8192, 1024, 1024], T=32768)
        run('o', 8192, [8192], T=32768)
        run('gate|up', 8192, [28672, 28672], T=32768)
        run('down', 28672, [8192], T=16384)
        sys.exit(0)
    run('q|k|v', 4096, [4096, 1024, 1024])
    run('o', 4096, [4096])
    run('gate|up', 4096, [14336, 14336])
    run('down', 14336, [4096])
