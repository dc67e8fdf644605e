"""k_fp8_cast alone (static scales: no absmax pass) on a 14336 x 4096 bf16 weight: the division-free form against FP8_EXACT_DIV,
both semantics, codes and fake values; and the absmax pass alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmc_amd import _ffi
if os.environ.get('LLMC_PROBE_LIB'):
    _ffi.LIB_PATH = os.environ['LLMC_PROBE_LIB']
    print('library', _ffi.LIB_PATH)

L = _ffi.lib()
w = torch.randn(14336, 4096, device='cuda') * 0.02
w[:, torch.randperm(4096, device='cuda')[:4]] *= 20.0          # outlier channels, like bench.py's synth_weight
w = w.to(torch.bfloat16)
n = w.numel()
s = (w.abs().max().float() / 448.0).reshape(1).contiguous()
codes = torch.empty(w.shape, dtype=torch.uint8, device='cuda')
fake = torch.empty_like(w)
ws = _ffi.workspace(L.llmc_fp8_quant_ws_bytes(1, n), w.device)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, mode in (('qtorch', 0x100), ('qtorch exact-div', 0x300), ('cast', 0), ('cast exact-div', 0x200)):
    for fk, out in ((0, codes), (1, fake)):
        us = timeit(lambda: L.llmc_fp8_quant(_ffi.ptr(w), _ffi.dt(w), 1, n, mode | fk, _ffi.ptr(out), _ffi.ptr(s), _ffi.dt(torch.float32), 1,
                                             None, _ffi.stream()))
        b = n * (3 if not fk else 4)
        print(f'{name:18s} {"fake " if fk else "codes"}  {us:7.1f} us  {b / us / 1e6:5.2f} TB/s (read once + write)')
sc = torch.empty(1, dtype=torch.float32, device='cuda')
us = timeit(lambda: L.llmc_fp8_quant(_ffi.ptr(w), _ffi.dt(w), 1, n, 0x100, _ffi.ptr(codes), _ffi.ptr(sc), _ffi.dt(torch.float32), 0,
                                     _ffi.ptr(ws), _ffi.stream()))
print(f'absmax + cast (dynamic, qtorch, codes) {us:7.1f} us')
amax = torch.empty(1, dtype=torch.bfloat16, device='cuda')
ws2 = _ffi.workspace(L.llmc_minmax_qparams_ws_bytes(1, n), w.device)
us = timeit(lambda: L.llmc_minmax_qparams(_ffi.ptr(w), _ffi.dt(w), 1, n, 1, 1, -1.0, 1.0, _ffi.ptr(amax), None, _ffi.ptr(ws2), _ffi.stream()))
print(f'absmax alone (llmc_minmax_qparams)     {us:7.1f} us  {2 * n / us / 1e6:5.2f} TB/s')
